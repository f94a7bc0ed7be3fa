set -e
mkdir -p gpurun_out
echo "== current library" | tee gpurun_out/diag_epin.log
python tools/diag_epin.py 2>&1 | tee -a gpurun_out/diag_epin.log
echo "== prev.so" | tee -a gpurun_out/diag_epin.log
DEI2I_LIB=$PWD/de-i2i-gan_amd/lib/prev.so python tools/diag_epin.py 2>&1 | tee -a gpurun_out/diag_epin.log
