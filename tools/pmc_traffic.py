"""Folds the rocprofv3 passes of tools/pmc_traffic.sh into pmc_traffic.json: per workload the dominant kernel, its
FETCH_SIZE / WRITE_SIZE (KiB, averaged over launches), HBM bytes per launch with the guide's gfx950 correction
(FETCH_SIZE under-reports coalesced reads by 2x, /opt/skills/guides/MI355X_MICROARCH.md; calibrated in round 1 on a
kernel that reads its input exactly once), the kernel time of the --kernel-trace pass and the hash of the kernel
sources."""
import collections
import csv
import glob
import hashlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
DOMINANT = ("k_spatial_lut", "k_spatial_u_lds", "k_spatial_u", "k_spatial_w", "k_transition_run", "k_transition_any")


def src_sha():
    h = hashlib.sha256()
    csrc = ROOT / "viewport-entropy-toolkit_amd" / "csrc"
    for f in sorted(csrc.glob("*.hpp")) + sorted(csrc.glob("*.hip")):
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def counter(root, workload, name):
    per_kernel = collections.defaultdict(list)
    for f in glob.glob(f"{root}/{workload}/{name}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and "vet::" in r["Kernel_Name"]:
                per_kernel[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return per_kernel


def main():
    root, workloads = sys.argv[1], sys.argv[2:]
    out = {"_note": "FETCH_SIZE/WRITE_SIZE in KiB per launch, separate --pmc passes; hbm_bytes_per_launch = "
                    "2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE correction); written by tools/pmc_traffic.sh"}
    sha = src_sha()
    for w in workloads:
        fetch, write = counter(root, w, "FETCH_SIZE"), counter(root, w, "WRITE_SIZE")
        stats = {}
        for f in glob.glob(f"{root}/{w}/trace/*/*_kernel_stats.csv"):
            for r in csv.DictReader(open(f)):
                stats[r["Name"].split("(")[0].replace("void ", "")] = (int(r["Calls"]), float(r["AverageNs"]), float(r["TotalDurationNs"]))
        cands = [k for k in stats if any(d in k for d in DOMINANT)]
        if not cands:
            out[w] = {"error": "no dominant kernel found", "kernels": sorted(stats)}
            continue
        kern = max(cands, key=lambda k: stats[k][2])           # the weighted workloads also run the unweighted variant
        if w in ("config3", "config4", "config2", "defaults"):
            lut = [k for k in cands if "k_spatial_lut" in k or "k_spatial_w" in k]
            kern = max(lut, key=lambda k: stats[k][2]) if lut else kern
        fk = sum(fetch.get(kern, [0])) / max(len(fetch.get(kern, [0])), 1)
        wk = sum(write.get(kern, [0])) / max(len(write.get(kern, [0])), 1)
        out[w] = {"kernel": kern, "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk,
                  "hbm_bytes_per_launch": 2 * fk * 1024 + wk * 1024,
                  "kernel_ms": stats[kern][1] / 1e6, "launches_in_trace": stats[kern][0], "kernel_src_sha": sha}
    Path(root, "pmc_traffic.json").write_text(json.dumps(out, indent=1))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
