"""Host-side track loader of the C-ABI (vet_csv_read_tracks) against pandas.read_csv, which is what
the reference parses user files with (utilities/data_utils.py:305-316): bit-identical FP64 columns
on everything the fast path accepts, VET_CSV_FALLBACK (-> pandas) on everything else.  CPU only."""

import os
import zlib

import numpy as np
import pandas as pd
import pytest

from viewport_entropy_toolkit import _ingest, _native
from viewport_entropy_toolkit.data_types import ValidationError

COLS = ["time", "2dmu", "2dmv"]


def pandas_columns(path):
    d = pd.read_csv(path, usecols=COLS)
    return [d[c].to_numpy(dtype=np.float64) for c in COLS]


def native_columns(path):
    (status, *cols), = _native.read_tracks([path], 1)
    return status, cols


def same_bits(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        return False
    nan = np.isnan(a)
    return bool(np.array_equal(nan, np.isnan(b)) and np.array_equal(a[~nan], b[~nan])
                and np.array_equal(np.signbit(a[~nan]) | (a[~nan] == 0), np.signbit(b[~nan]) | (b[~nan] == 0)))


def write(tmp_path, text, name="u.csv", newline="\n"):
    p = tmp_path / name
    p.write_bytes(text.replace("\n", newline).encode())
    return p


FORMATS = ["%.6f", "%.12f", "%.17g", "%r", "%.3e", "%.15E", "%g", "%.20f", "%.1f"]


@pytest.mark.parametrize("fmt", FORMATS)
@pytest.mark.parametrize("newline", ["\n", "\r\n"])
def test_numeric_files_match_pandas_bit_for_bit(tmp_path, fmt, newline):
    rng = np.random.default_rng(zlib.crc32((fmt + newline).encode()))
    n = 400
    t = np.cumsum(rng.random(n) * 0.2)
    mu, mv = rng.random(n), rng.random(n)
    mu[::37] = 0.0
    mv[::41] = 1.0
    f = (lambda v: repr(float(v))) if fmt == "%r" else (lambda v: fmt % v)
    lines = ["extra,2dmv,time,junk,2dmu"]               # column order differs from usecols; extra columns
    for i in range(n):
        lines.append(f"{i},{f(mv[i])},{f(t[i])},x{i},{f(mu[i])}")
    p = write(tmp_path, "\n".join(lines) + "\n", newline=newline)
    status, cols = native_columns(p)
    assert status == _native.VET_CSV_OK
    for got, ref in zip(cols, pandas_columns(p)):
        assert same_bits(got, ref)


def test_random_decimal_strings_convert_like_pandas(tmp_path):
    """The converter itself: 60 000 random decimal spellings with 1-25 digits, signs, exponents (integer
    parts of at most 18 digits: beyond 64 bits pandas' dtype inference may keep the column as text, and the
    loader hands such files to pandas — test_integer_parts_beyond_64_bits_are_left_to_pandas)."""
    rng = np.random.default_rng(99)
    toks = []
    for _ in range(60000):
        nd = int(rng.integers(1, 26))
        digits = "".join(rng.choice(list("0123456789"), nd))
        cut = int(rng.integers(0, min(nd, 18) + 1))
        s = digits[:cut] + "." + digits[cut:]
        if s == ".":
            s = "0."
        kind = rng.integers(0, 6)
        if kind == 0:
            s = "-" + s
        elif kind == 1:
            s = "+" + s
        if rng.integers(0, 4) == 0:
            s += rng.choice(["e", "E"]) + rng.choice(["", "+", "-"]) + str(int(rng.integers(0, 40)))
        toks.append(s)
    text = "time,2dmu,2dmv\n" + "\n".join(f"{a},{b},{c}" for a, b, c in zip(toks[0::3], toks[1::3], toks[2::3])) + "\n"
    p = write(tmp_path, text)
    status, cols = native_columns(p)
    assert status == _native.VET_CSV_OK
    for got, ref in zip(cols, pandas_columns(p)):
        assert same_bits(got, ref)


@pytest.mark.parametrize("tok,ok", [
    ("18446744073709551615e0", True), ("18446744073709551616e0", False), ("99999999999999999999.5", False),
    ("-9223372036854775808.5", True), ("-9223372036854775809.5", False), ("0.5\n111111111111111111111.0", False),
    ("00018446744073709551616.5", False), ("+9999999999999999999.5", True), ("0." + "9" * 30, True)])
def test_integer_parts_beyond_64_bits_are_left_to_pandas(tmp_path, tok, ok):
    """pandas tries int64, uint64, then float64 per column; an integer part that overflows both makes it keep the
    column as text (found by the byte-level fuzz, tests/test_csv_loader_asan.py).  The fast path never answers for
    such a file; what it does accept equals pandas bit for bit."""
    p = write(tmp_path, "time,2dmu,2dmv\n" + "\n".join(f"{t},0.5,0.5" for t in tok.split("\n")) + "\n")
    status, cols = native_columns(p)
    assert (status == _native.VET_CSV_OK) == ok
    if ok:
        ref = pd.read_csv(p)
        assert ref["time"].dtype == np.float64 and same_bits(cols[0], ref["time"].to_numpy())


def test_missing_values_short_rows_blank_lines_and_bom(tmp_path):
    text = ("﻿time,2dmu,2dmv,other\n"
            "0.0,0.5,0.25,a\n"
            "\n"
            "0.1,,0.5,b\n"
            "0.2,NaN,0.5,c\n"
            "0.3,0.5,NA,d\n"
            "0.4,0.25\n"                       # short row: 2dmv missing
            "0.5,0.5,null,\n"
            "0.6,0.75,0.125,e\n"
            "7,1,0,f")                          # integers, no trailing newline
    p = write(tmp_path, text)
    status, cols = native_columns(p)
    assert status == _native.VET_CSV_OK
    for got, ref in zip(cols, pandas_columns(p)):
        assert same_bits(got, ref)
    assert np.isnan(cols[1]).sum() == 2 and np.isnan(cols[2]).sum() == 3


@pytest.mark.parametrize("text", [
    'time,2dmu,2dmv\n0.0,"0.5",0.5\n',                  # quoted field
    "time,2dmu,2dmv\n0.0,abc,0.5\n",                    # text
    "time,2dmu,2dmv\n0.0,inf,0.5\n",                    # infinity spelling
    "time,2dmu,2dmv\n0.0,0.5,0.5,9\n",                  # more fields than the header (pandas infers an index)
    "time,2dmu\n0.0,0.5\n",                             # column missing
    "time,2dmu,2dmv,time\n0.0,0.5,0.5,1\n",             # duplicate name
    "time,2dmu,2dmv\n0.0, 0.5,0.5\n",                   # padded field
    "time,2dmu,2dmv\n123456789012345678,0.5,0.5\n",     # integer beyond FP64's exact range
    "time,2dmu,2dmv\n0.0,1e400,0.5\n",                  # overflow
    "",                                                 # empty file
])
def test_everything_else_is_left_to_pandas(tmp_path, text):
    p = write(tmp_path, text)
    status, _ = native_columns(p)
    assert status == _native.VET_CSV_FALLBACK
    # and the directory reader then behaves exactly like the pandas-only path
    def run(mode):
        os.environ["VET_CSV_PARSER"] = mode
        try:
            return ("ok", _ingest.read_directory([p], 100, 200))
        except ValidationError as e:
            return ("error", str(e))
        finally:
            os.environ.pop("VET_CSV_PARSER", None)
    a, b = run("native"), run("pandas")
    assert a[0] == b[0]
    if a[0] == "error":
        assert a[1] == b[1]
    else:
        for x, y in zip(a[1][0][:4], b[1][0][:4]):
            assert np.array_equal(x, y, equal_nan=True)


def test_missing_file_reports_io(tmp_path):
    (status, *_), = _native.read_tracks([tmp_path / "nope.csv"], 1)
    assert status == _native.VET_CSV_IO
    with pytest.raises(ValidationError, match="File not found"):
        _ingest.read_directory([tmp_path / "nope.csv"], 100, 200)


def test_directory_reader_equals_pandas_path_on_ragged_tracks(tmp_path):
    rng = np.random.default_rng(5)
    files = []
    for u in range(20):
        n = int(rng.integers(50, 300))
        t = np.round(np.cumsum(rng.random(n) * 0.15), 3) + 3.0
        rows = [f"{float(t[i])!r},{float(rng.random())!r},{float(rng.random())!r}" for i in range(n)]
        for j in rng.integers(0, n, 5):
            rows[j] = f"{float(t[j])!r},,{float(rng.random())!r}"      # dropped by dropna
        files.append(write(tmp_path, "time,2dmu,2dmv\n" + "\n".join(rows) + "\n", name=f"user{u:02d}.csv"))
    os.environ["VET_CSV_PARSER"] = "pandas"
    try:
        ref = _ingest.read_directory(files, 100, 200)
    finally:
        os.environ.pop("VET_CSV_PARSER")
    got = _ingest.read_directory(files, 100, 200, threads=4)
    assert len(got) == len(ref)
    for g, r in zip(got, ref):
        assert g[4] == r[4]
        for x, y in zip(g[:4], r[:4]):
            assert np.array_equal(x, y)
