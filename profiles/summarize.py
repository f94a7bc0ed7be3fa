"""Turn the rocprofv3 outputs of profiles/collect.sh into the committed per-round summaries.

  python profiles/summarize.py <tag> <gpurun_out dir>     e.g.  python profiles/summarize.py r01_d gpurun_out

Reads   <dir>/prof_<tag>/**/_kernel_stats.csv           (rocprofv3 --kernel-trace --stats)
        <dir>/pmc_fetch_<tag>/**/_counter_collection.csv (rocprofv3 --pmc FETCH_SIZE, own pass)
        <dir>/pmc_write_<tag>/**/_counter_collection.csv (rocprofv3 --pmc WRITE_SIZE, own pass)
Writes  profiles/<tag>_kernel_stats.csv   (copy of the --stats table)
        profiles/<tag>_pmc_traffic.json   per kernel family: launches, HBM bytes per launch
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB, and gfx950's FETCH_SIZE tallies 128-byte
requests at 64 bytes (MI355X_MICROARCH.md, HBM / rocprofv3 section), so it is doubled; WRITE_SIZE is used as read."""
import csv
import glob
import json
import shutil
import sys
from collections import defaultdict
from pathlib import Path

FAMILIES = (("thin_cin_conv", "thin_cin_conv_kernel"), ("thin_cout_conv", "thin_cout_conv_kernel"), ("halo16_conv", "halo16_conv_kernel"), ("halo_conv", "halo_conv_kernel"), ("gather_gemm_v2", "gather_gemm_v2_kernel"), ("gather_gemm_v1", "gather_gemm_kernel"),
            ("wgrad_halo", "wgrad_halo_kernel"), ("wgrad_thin", "wgrad_thin_kernel"), ("wgrad_v2", "wgrad_v2_kernel"), ("wgrad_reduce_unpack", "wgrad_reduce_unpack_kernel"),
            ("wgrad_v1", "wgrad_kernel"))


def family(name):
    # the two large instances of the 16 x 32-tile halo conv are kernels of their own: the pipelined forward instance (the
    # step's largest kernel, bench.py's roofline line) and the FOLD instance (reflect dgrads)
    if "halo16_conv_kernel<128, 8, 0, false, 0, true" in name:
        return "halo16_conv_fwd"
    if "halo16_conv_kernel<128, 8, 0, true" in name:
        return "halo16_conv_fold"
    if "halo16_conv_kernel<128, 4, 0, false, 0, false, false, true" in name:
        return "halo16_conv_s2"         # the stride-2 4x4 form (parity planes in LDS)
    for fam, key in FAMILIES:
        if key + "<" in name or key + "(" in name:
            return fam
    return None


def counter(pattern, cname):
    per = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(pattern, recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != cname:
                    continue
                fam = family(r["Kernel_Name"])
                if fam:
                    per[fam][0] += 1
                    per[fam][1] += float(r["Counter_Value"])
    return per


def main():
    tag, root = sys.argv[1], Path(sys.argv[2])
    out = Path(__file__).resolve().parent
    stats = glob.glob(str(root / f"prof_{tag}" / "**" / "*_kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], out / f"{tag}_kernel_stats.csv")
    fetch = counter(str(root / f"pmc_fetch_{tag}" / "**" / "*_counter_collection.csv"), "FETCH_SIZE")
    write = counter(str(root / f"pmc_write_{tag}" / "**" / "*_counter_collection.csv"), "WRITE_SIZE")
    res = {"_units": "FETCH_SIZE/WRITE_SIZE in KiB per launch (averages); hbm_bytes_per_launch = (2*FETCH + WRITE)*1024",
           "_command": "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline (bf16, 256x256, batch 16)"}
    for fam in sorted(set(fetch) | set(write)):
        nf, sf = fetch.get(fam, (0, 0.0))
        nw, sw = write.get(fam, (0, 0.0))
        if not nf or not nw:
            continue
        res[fam] = {"launches": nf, "FETCH_SIZE_KB_avg": sf / nf, "WRITE_SIZE_KB_avg": sw / nw,
                    "hbm_bytes_per_launch": (2 * sf / nf + sw / nw) * 1024}
    if len(res) > 2:
        (out / f"{tag}_pmc_traffic.json").write_text(json.dumps(res, indent=1))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
