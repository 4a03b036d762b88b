"""The C-ABI library loads and exports every symbol include/vet.h declares; the ctypes table in
_native matches the header.  No compute calls (runs without a GPU)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "vet.h"


def header_functions():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(vet_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    fns = header_functions()
    for must in ("vet_create", "vet_plan_create", "vet_spatial_entropy", "vet_transition_entropy",
                 "vet_spatial_entropy_ids", "vet_transition_entropy_ids", "vet_last_error"):
        assert must in fns


def test_library_exports_every_declared_symbol():
    from viewport_entropy_toolkit import _native
    lib = ctypes.CDLL(str(_native.LIB_PATH))          # fails loudly if the extension is not built
    for name in header_functions():
        assert hasattr(lib, name), f"{name} declared in include/vet.h but not exported"


def test_ctypes_table_matches_header():
    from viewport_entropy_toolkit import _native
    assert sorted(_native.SIGNATURES) == header_functions()
    lib = _native.load_library()
    assert lib.vet_version() == 141
    assert [lib.vet_kernel_name(i).decode() for i in range(8)] == \
        ["k_grid_dirs", "k_nearest_lut", "k_spatial", "k_transition", "k_finalize", "k_wtab", "k_weights", ""]
    assert _native.KERNEL_IDS == {lib.vet_kernel_name(i).decode(): i for i in range(7)}


def test_no_cpu_fallback_without_device():
    """Without a GPU the engine must refuse to start instead of computing somewhere else."""
    from viewport_entropy_toolkit import _native
    lib = _native.load_library()
    if lib.vet_device_count() > 0:
        pytest.skip("a GPU is visible; the refusal path is for GPU-less hosts")
    with pytest.raises(_native.NativeUnavailable):
        _native.Engine(0)
    from viewport_entropy_toolkit.utilities import compute_spatial_entropy, generate_fibonacci_lattice, EntropyConfig
    from viewport_entropy_toolkit import Vector
    with pytest.raises(_native.NativeUnavailable):
        compute_spatial_entropy({"a": Vector(1.0, 0.0, 0.0)}, generate_fibonacci_lattice(20), EntropyConfig())
    from viewport_entropy_toolkit.utilities import vector_angle_distance, find_angular_distances
    with pytest.raises(_native.NativeUnavailable):          # not wrapped into the reference's ValidationError
        vector_angle_distance(Vector(1.0, 0.0, 0.0), Vector(0.0, 1.0, 0.0))
    with pytest.raises(_native.NativeUnavailable):
        find_angular_distances(Vector(1.0, 0.0, 0.0), generate_fibonacci_lattice(20))


def test_product_does_not_import_the_oracle():
    """oracle/ is test infrastructure: nothing shipped may import, load or execute it
    (comments may cite it)."""
    pkg = ROOT / "viewport-entropy-toolkit_amd"
    py_pat = re.compile(r"^\s*(from|import)\s+oracle\b|[\"']oracle[/\"']|libvet_oracle", re.M)
    for f in pkg.rglob("*.py"):
        assert not py_pat.search(f.read_text()), f"{f} references the oracle"
    for f in list(pkg.rglob("*.hip")) + list(pkg.rglob("*.hpp")) + list(pkg.rglob("Makefile")):
        code = re.sub(r"//.*|#.*", "", f.read_text())
        assert "oracle" not in code, f"{f} references the oracle"


def test_one_hip_runtime_is_loaded():
    """_native.load_library() preloads the HIP runtime the installed torch ships (same SONAME as the system one), so the
    process ends up with exactly one libamdhip64 whatever is imported later ('One HIP runtime per process').
    Checked in a child process: this one may have loaded things already."""
    import subprocess
    import sys
    code = (
        "import sys, re\n"
        f"sys.path.insert(0, {str(ROOT / 'viewport-entropy-toolkit_amd')!r})\n"
        "from viewport_entropy_toolkit import _native\n"
        "_native.load_library()\n"
        "import torch\n"
        "libs = sorted(set(re.findall(r'\\S*libamdhip64\\S*', open('/proc/self/maps').read())))\n"
        "print(len(libs), _native.HIP_RUNTIME_PRELOADED in libs if _native.HIP_RUNTIME_PRELOADED else 'system')\n"
    )
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-1500:]
    n, ok = out.stdout.split()[-2:]
    assert n == "1" and ok in ("True", "system"), out.stdout


def test_environment_is_read_in_one_place_only():
    """The tuning environment is parsed once, in vet_create (vet_context.hip): no launch path may call getenv, and nothing
    experimental (tools/experiments/) is compiled into the library."""
    csrc = ROOT / "viewport-entropy-toolkit_amd" / "csrc"
    for f in list(csrc.glob("*.hip")) + list(csrc.glob("*.hpp")):
        text = f.read_text()
        code = "\n".join(line.split("//")[0] for line in text.splitlines())
        if f.name != "vet_context.hip":
            assert "getenv" not in code, f"{f.name} reads the environment"
        assert "experiments/" not in code and "k_spatial_rows" not in code and "k_spatial_walk" not in code, f.name
    mk = (csrc / "Makefile").read_text()
    assert "experiments" not in mk
    # timing-only switches (wrong results by design) live in tools/experiments/ as a patch, not in the product sources
    for f in list(csrc.glob("*")):
        if f.is_file():
            assert "VET_EXP" not in f.read_text(), f"{f.name} holds a timing-only experiment switch"
    product_flags = [line for line in mk.splitlines() if line.startswith("CXXFLAGS") and "VFLAGS" not in line]
    assert product_flags


def test_policy_constant_matches_the_header():
    from viewport_entropy_toolkit import _native
    m = re.search(r"#define VET_TABLE_SAMPLES_PER_DIRECTION (\d+)", HEADER.read_text())
    assert m and int(m.group(1)) == _native.TABLE_SAMPLES_PER_DIRECTION
