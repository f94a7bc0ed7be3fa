#!/usr/bin/env python3
"""LDS tile-size sweep of the convolution kernels on the train step (BASELINE.json configs[3]: the 512x512 generator with
one more encoder / decoder scale; the 256x256 batch-16 headline configuration beside it).

Run ON THE GPU BOX from the repo root:   python3 profiles/sweep_tiles.py [tag]

Each row is one `bench.py` run (child process: 2 warm-up + 5 timed steps, HIP-event roofline leg on) with the library's
tile options set through `--set-option`:
  halo_bn / halo_stages : output-channel width of the halo conv's 256-pixel tile (64 | 128) and the depth of its weight
                          ring (LDS = 88,064 B halo + stages x bn x 128 B; 4 is the
                          shallowest ring the two-group schedule is correct for -- see conv_halo.hip)
  halo_conv = 0         : no halo kernel -- the LDS-DMA gather GEMM (256x128 / 192x128 / 256x64 tiles) takes those layers
  + gather_gemm_v2 = 0  : ... and without that one too: 128x128 register-staged tiles everywhere
  wgrad_halo = 0        : weight gradients of the 3x3 layers through the gather wgrad (128x128 tiles) instead of the
                          halo wgrad (128 co x 64 ci register blocks per 9 taps)
Writes gpurun_out/sweep_tiles_<tag>.json (rows) and prints a table; the committed copy lives under profiles/."""
import json
import os
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
ROWS = [
    ("shipped (bn 128|64 by layer, 4-deep ring)", []),
    ("halo bn=64 everywhere, 4-deep ring", ["halo_bn=64", "halo_stages=4"]),
    ("halo bn=64 everywhere, 6-deep ring", ["halo_bn=64", "halo_stages=6"]),
    ("halo bn=64 everywhere, 8-deep ring", ["halo_bn=64", "halo_stages=8"]),
    ("no halo conv (LDS-DMA gather GEMM tiles)", ["halo_conv=0"]),
    ("no halo conv, no LDS-DMA gather (128x128 tiles)", ["halo_conv=0", "gather_gemm_v2=0"]),
    ("no halo wgrad (gather wgrad 128x128)", ["wgrad_halo=0"]),
]
SIZES = [(512, 4), (256, 16)]


def run(size, batch, options):
    cmd = [sys.executable, str(REPO / "bench.py"), "--image-size", str(size), "--batch", str(batch), "--steps", "5", "--warmup", "2",
           "--no-cpu-baseline"]
    for o in options:
        cmd += ["--set-option", o]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if out.returncode != 0 or not lines:
        return {"error": (out.stderr or out.stdout)[-400:]}
    return json.loads(lines[-1])


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    rows = []
    for size, batch in SIZES:
        for name, options in ROWS:
            j = run(size, batch, options)
            row = {"image_size": size, "batch": batch, "variant": name, "options": options}
            if "error" in j:
                row["error"] = j["error"]
            else:
                r, m = j["roofline"], j["mfma"]
                row.update(ms_per_step=j["ms_per_step"], pairs_per_s=j["value"], halo_conv_tflops=r["achieved"],
                           halo_conv_launches_per_step=r["launches_per_step"], halo_conv_avg_launch_ms=r["avg_launch_ms"],
                           conv_fwd_dgrad_tflops=r["all_conv_fwd_dgrad_kernels"]["achieved"], wgrad_tflops=m["wgrad_tflops"],
                           step_mfma_util=m["step_mfma_util"])
            rows.append(row)
            print("%4d b%-2d  %-50s %s" % (size, batch, name,
                                           row.get("error", "")[:80] if "error" in row else
                                           "%7.2f ms/step  halo %6.0f TF/s  fwd+dgrad %6.0f TF/s  wgrad %6.0f TF/s" %
                                           (row["ms_per_step"], row["halo_conv_tflops"], row["conv_fwd_dgrad_tflops"], row["wgrad_tflops"])),
                  flush=True)
    out = REPO / "gpurun_out"
    os.makedirs(out, exist_ok=True)
    with open(out / f"sweep_tiles_{tag}.json", "w") as f:
        json.dump({"command": "python3 profiles/sweep_tiles.py", "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
