# wall time of the halo16 kernel (res-block shape) with parts of its loop removed: 10 + mask (1 no DMA, 2 no reads, 4 no MFMA)
for a in 6 11 12 14 13 15 16 17; do echo -n "ablate $a: "; python tools/diag_halo16_stamps.py 256 256 64 16 1 $a 0 2>&1 | grep -E "plain"; done
