"""profiles/<tag>_pmc_mfma.json from the MFMA-busy PMC pass of profiles/collect.sh (or mfma_only.sh):
    python profiles/summarize_mfma.py <tag> gpurun_out
Per kernel family: fraction of SIMD-cycles in which the MFMA pipe was busy =
    SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8) * 1024)
SQ_VALU_MFMA_BUSY_CYCLES sums real cycles over the 1024 SIMDs (calibration on the halo conv: 9.2e7 per launch = 16 cycles x
the 6.0e6 v_mfma_f32_16x16x32_bf16 of a launch, i.e. the dense rate of one such MFMA per 16 cycles per SIMD);
GRBM_GUI_ACTIVE accumulates over the 8 XCDs (18.7 counts per ns of kernel time = 8 x 2.33 GHz), hence the / 8 -- which also
gives the shader clock held during each family's launches."""
import csv
import glob
import json
import sys
from collections import defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from summarize import family  # noqa: E402

XCDS, SIMDS = 8, 1024


def main():
    tag, root = sys.argv[1], Path(sys.argv[2])
    f = glob.glob(str(root / f"pmc_mfma_{tag}" / "**" / "*_counter_collection.csv"), recursive=True)[0]
    busy, gui, ns, n = defaultdict(float), defaultdict(float), defaultdict(float), defaultdict(int)
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            fam = family(r["Kernel_Name"]) or "non_mfma_kernels"
            if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
                busy[fam] += float(r["Counter_Value"])
            elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                gui[fam] += float(r["Counter_Value"])
                ns[fam] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                n[fam] += 1
    out = {"_units": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8) * 1024): fraction of SIMD-cycles with the "
                     "MFMA pipe busy while the family's kernels ran; clock_ghz = GRBM_GUI_ACTIVE / 8 / kernel ns (meaningful for the "
                     "long kernels only: the counter window of a short launch is wider than its begin-end timestamps)",
           "_command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 2 --warmup 1 "
                       "--no-cpu-baseline --no-roofline (bf16, 256x256, batch 16); own pass, no other trace domain"}
    for fam in sorted(gui, key=lambda k: -busy[k]):
        cyc = gui[fam] / XCDS
        out[fam] = {"launches": n[fam], "mfma_busy_frac": busy[fam] / (cyc * SIMDS) if cyc else None,
                    "clock_ghz": cyc / ns[fam] if ns[fam] else None, "kernel_ms_total": ns[fam] / 1e6}
    tb, tg = sum(busy.values()), sum(gui.values()) / XCDS
    out["all_kernels_of_the_run"] = {"mfma_busy_frac": tb / (tg * SIMDS)}
    with open(Path(__file__).resolve().parent / f"{tag}_pmc_mfma.json", "w") as fh:
        json.dump(out, fh, indent=1)
    for k, v in out.items():
        if not k.startswith("_"):
            print(k, v)


if __name__ == "__main__":
    main()
