"""Folds the summaries of tools/pmc_sq.sh (gpurun_out/<tag>/summary.json, one tag per workload) into profiles/pmc_sq.json:
per workload the dominant kernel's SQ counters per launch, with the hash of the kernel sources they were collected on
(bench.py's `roofline.secondary` of the transition workloads reads it and drops a stale record).
usage: python tools/pmc_sq_record.py <workload>=<pmc_sq summary.json>[+<pmc3 summary.json>] ..."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
from pmc_traffic import src_sha  # noqa: E402

DOMINANT = {"config5": "k_transition", "config4": "k_spatial_lut", "config3": "k_spatial_lut", "config2": "k_spatial_lut",
            "defaults": "k_spatial_lut"}


def main():
    out_f = ROOT / "profiles" / "pmc_sq.json"
    out = json.loads(out_f.read_text()) if out_f.exists() else {}
    out["_note"] = ("SQ counters per launch of the workload's dominant kernel (rocprofv3 --pmc passes of tools/pmc_sq.sh, four "
                    "counters per pass, averaged over the launches of a pass); written by tools/pmc_sq_record.py")
    sha = src_sha()
    for arg in sys.argv[1:]:
        w, f = arg.split("=", 1)
        f, _, extra = f.partition("+")           # workload=pmc_sq summary[+pmc3 (stall / L2 request) summary of the same sources]
        summ = json.loads(Path(f).read_text())
        want = DOMINANT.get(w, "k_")
        kerns = [k for k in summ if k != "kernel_stats" and (want in k or want[2:] in k)]
        if not kerns:
            kerns = [k for k in summ if k != "kernel_stats"]
        k = max(kerns, key=lambda q: summ[q].get("SQ_BUSY_CYCLES", 0))
        out[w] = dict(summ[k], kernel=k, kernel_src_sha=sha)
        if extra:
            st = json.loads(Path(extra).read_text())
            ks = [q for q in st if q != "kernel_stats" and (want in q or want[2:] in q)] or [q for q in st if q != "kernel_stats"]
            q = max(ks, key=lambda z: st[z].get("TCC_REQ_sum", 0))
            for name in ("TCC_REQ_sum", "TCC_READ_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_GATE_EN1_sum", "TD_TC_STALL_sum"):
                if name in st[q]:
                    out[w][name] = st[q][name]
    out_f.write_text(json.dumps(out, indent=1))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
