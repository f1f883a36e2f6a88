# usage: tools/cmp_variants.sh lib1.so lib2.so ...   (full-iteration kernel time of each build, 1e6 protons)
mkdir -p gpurun_out/var
for L in "$@"; do
  echo "== $L" >> gpurun_out/var/res.log
  MCS_HIP_LIB=$L timeout -k 10 200 python tools/gpu_run.py 1000000 45 2>&1 | tail -1 >> gpurun_out/var/res.log
done
cat gpurun_out/var/res.log
