# A/B of two builds of the library on the bench scene (tools/gpu_t4_probe.py in separate processes, alternating): FTN_LIB picks the .so
mkdir -p gpurun_out/ab
for i in 1 2; do
  for lib in libfountain_hip_base.so libfountain_hip.so; do
    echo "== $lib" >> gpurun_out/ab/ab.log
    FTN_LIB=$lib timeout -k 10 200 python tools/gpu_t4_probe.py -- '' 2>&1 | grep '^{' >> gpurun_out/ab/ab.log || exit 1
  done
done
cat gpurun_out/ab/ab.log
