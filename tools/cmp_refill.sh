mkdir -p gpurun_out/var; rm -f gpurun_out/var/refill.log
for L in libmcs_hip_r2.so libmcs_hip_r4.so libmcs_hip_r8.so libmcs_hip_r16.so; do
  echo "== $L" >> gpurun_out/var/refill.log
  MCS_HIP_LIB=$L timeout -k 10 200 python tools/gpu_run.py 1000000 45 2>&1 | grep -E "pcut  (1|5) |pcut (17|22|26|30) |TOTAL" >> gpurun_out/var/refill.log
done
cat gpurun_out/var/refill.log
