"""Sanitizer + fuzz leg of the host-side track loader (csrc/vet_ingest.cpp): the loader alone is built with
AddressSanitizer + UndefinedBehaviorSanitizer (`make -C viewport-entropy-toolkit_amd/csrc asan`, host code
only) into a driver executable; the regression corpus of test_csv_loader.py and hypothesis-made byte soup
(truncated files, lone CR, huge digit strings, absurd exponents, NUL bytes, quotes, ragged rows) go through
it.  A sanitizer report aborts the driver; its per-file answer (status, rows, hash of the three FP64 columns)
must equal the production library's, which test_csv_loader.py pins to pandas bit for bit.  CPU only."""
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from viewport_entropy_toolkit import _native

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "viewport-entropy-toolkit_amd" / "csrc"
DRIVER = ROOT / "viewport-entropy-toolkit_amd" / "lib" / "vet_ingest_asan"


@pytest.fixture(scope="module")
def driver():
    if not shutil.which("g++"):
        pytest.skip("no host compiler")
    subprocess.run(["make", "-C", str(CSRC), "asan"], check=True, capture_output=True)
    assert DRIVER.exists()
    return DRIVER


def fnv(arrays):
    h = 0xCBF29CE484222325
    for a in arrays:
        for byte in np.ascontiguousarray(a, dtype=np.float64).tobytes():
            h = ((h ^ byte) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def check(driver, paths):
    """Run the sanitizer build over ``paths`` and compare with the production library, file by file."""
    out = subprocess.run([str(driver)] + [str(p) for p in paths], capture_output=True, text=True, timeout=300,
                         env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert out.returncode == 0, f"sanitizer build failed (rc {out.returncode}):\n{out.stderr[-3000:]}"
    lines = out.stdout.strip().splitlines()
    assert len(lines) == len(paths)
    prod = _native.read_tracks(paths, 2)
    for line, (status, *cols), p in zip(lines, prod, paths):
        s, n, h = line.split()
        assert int(s) == status, p
        if status == _native.VET_CSV_OK:
            assert int(n) == len(cols[0]), p
            assert int(h, 16) == fnv(cols), p
    return prod


def test_regression_corpus_under_sanitizers(tmp_path, driver):
    rng = np.random.default_rng(7)
    files = []

    def add(text, newline="\n", raw=None):
        p = tmp_path / f"f{len(files):03d}.csv"
        p.write_bytes(raw if raw is not None else text.replace("\n", newline).encode())
        files.append(p)

    for fmt in ["%.6f", "%.17g", "%.3e", "%.20f", "%g"]:
        for nl in ("\n", "\r\n", "\r"):
            rows = [f"{i},{fmt % rng.random()},{fmt % (i * 0.1)},x,{fmt % rng.random()}" for i in range(200)]
            add("extra,2dmv,time,junk,2dmu\n" + "\n".join(rows) + "\n", nl)
    add("﻿time,2dmu,2dmv,other\n0.0,0.5,0.25,a\n\n0.1,,0.5,b\n0.2,NaN,0.5,c\n0.4,0.25\n7,1,0,f")
    for text in ['time,2dmu,2dmv\n0.0,"0.5",0.5\n', "time,2dmu,2dmv\n0.0,abc,0.5\n", "time,2dmu,2dmv\n0.0,inf,0.5\n",
                 "time,2dmu,2dmv\n0.0,0.5,0.5,9\n", "time,2dmu\n0.0,0.5\n", "time,2dmu,2dmv,time\n0,0,0,1\n",
                 "time,2dmu,2dmv\n0.0, 0.5,0.5\n", "time,2dmu,2dmv\n123456789012345678,0.5,0.5\n",
                 "time,2dmu,2dmv\n0.0,1e400,0.5\n", "", "\n\n\n", "time,2dmu,2dmv", "time,2dmu,2dmv\n", ",,,\n,,,\n",
                 "time,2dmu,2dmv\n" + "9" * 1000000 + ",0.5,0.5\n",            # a million-digit field
                 "time,2dmu,2dmv\n0." + "3" * 1000000 + ",0.5,0.5\n",
                 "time,2dmu,2dmv\n1e+99999,1e-99999,1E99999999999999999999\n",
                 "time,2dmu,2dmv\n1e,1e+,1e-\n", "time,2dmu,2dmv\n.,-,+\n", "time,2dmu,2dmv\n-.5,+.5,5.\n",
                 "time,2dmu,2dmv\n0.5\r0.25,1\r\r\n\r", "time,2dmu,2dmv\n" + "," * 100000 + "\n"]:
        add(text)
    add(None, raw=b"time,2dmu,2dmv\n0.1,0.2\x00,0.3\n")
    add(None, raw=b"\xef\xbb")                                                   # truncated BOM
    add(None, raw=b"\xef\xbb\xbftime,2dmu,2dmv\n0.5,0.5,0.5")
    add(None, raw=bytes(rng.integers(0, 256, 4096, dtype=np.uint8)))
    files.append(tmp_path / "does_not_exist.csv")
    prod = check(driver, files)
    assert sum(1 for s, *_ in prod if s == _native.VET_CSV_OK) >= 15


# bytes a CSV of numbers is made of, weighted toward the structural ones
ALPHABET = b"0123456789" * 3 + b".,,\n\n\r-+eE" + b" \tNaN\"x\x00"
digits = st.text("0123456789", min_size=1, max_size=22)
token = st.one_of(
    st.builds(lambda s, a, b: s + a + "." + b, st.sampled_from(["", "", "-", "+"]), digits, digits),
    st.builds(lambda a, e: a + "e" + e, digits, st.sampled_from(["0", "5", "-3", "+12", "-40", "99", "-101", "308", "99999"])),
    digits.map(lambda d: d[:15]), digits.map(lambda d: "." + d), digits.map(lambda d: d + "."),
    st.sampled_from(["", "", "NaN", "nan", "NA", "null", "#N/A", "-NaN", "0", "1", "1.0", "0.5", "-0.0", "1e0", "inf", "abc", " 1"]))
row = st.lists(token, min_size=1, max_size=5).map(lambda t: (",".join(t) + "\n").encode())
noise = st.binary(min_size=0, max_size=12).map(lambda b: bytes(ALPHABET[c % len(ALPHABET)] for c in b))
fragment = st.one_of(row, row, row, row, row, row, noise,
                     st.sampled_from([b"\r", b"\r\n", b"\n", b",,\n", b"9" * 400 + b",1,1\n", b"0." + b"1" * 400 + b",0,0\n",
                                      b"\xef\xbb\xbf", b"time,2dmu,2dmv\n", b"\"", b"\x00"]))


@settings(max_examples=int(__import__("os").environ.get("VET_FUZZ_EXAMPLES", "300")), deadline=None, derandomize="VET_FUZZ_EXAMPLES" not in __import__("os").environ, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(st.lists(st.lists(fragment, min_size=0, max_size=30).map(b"".join), min_size=8, max_size=8))
def test_byte_soup_under_sanitizers(tmp_path_factory, driver, blobs):
    """8 files per driver run: random concatenations of CSV-ish fragments, half of them behind a valid header,
    each also cut at a random point (truncated writes)."""
    d = tmp_path_factory.mktemp("soup")
    paths = []
    for i, blob in enumerate(blobs):
        header = [b"time,2dmu,2dmv\n", b"2dmv,x,time,2dmu\n", b"time,2dmu,2dmv\r\n", b""][i % 4] if i < 6 else b"time,2dmu,2dmv\n"
        data = header + blob
        if i >= 6 and data:                                       # truncated write
            data = data[: (len(blob) * 7919 + i) % (len(data) + 1)]
        p = d / f"s{i}.csv"
        p.write_bytes(data)
        paths.append(p)
    prod = check(driver, paths)
    # whatever the fast path accepts, pandas (the reference's parser, utilities/data_utils.py:314) reads the same bits
    import pandas as pd
    for (status, *cols), p in zip(prod, paths):
        if status != _native.VET_CSV_OK:
            continue
        ref = pd.read_csv(p, usecols=["time", "2dmu", "2dmv"])
        for got, name in zip(cols, ["time", "2dmu", "2dmv"]):
            want = ref[name].to_numpy(dtype=np.float64)
            assert got.shape == want.shape, (p.read_bytes(), got, want)
            both = np.isnan(got) & np.isnan(want)
            assert np.array_equal(got[~both], want[~both]), (p.read_bytes(), got, want)
