# kernel statistics of one tile-serial frame on the queues (config-5 scene, 1 spp)
set -e
ROOT=$PWD
mkdir -p gpurun_out/serial
cd /tmp && export TMPDIR=/tmp
cat > /tmp/serial_once.py <<PY
import sys
sys.path.insert(0, "$ROOT")
from fountain_amd import *
from fountain_amd import scenes, _abi as A
gpu = default_backend()
b, cam, r = scenes.instanced_cubes(gpu, n_copies=2309, res=(4096, 4096)); sc = b.create_scene()
si = SamplerIntegrator(cam, PathIntegrator(5, 1.0))
st = si.render_parallel(sc, Film(gpu, r), RandomSampler(1, 0), pipeline=A.FTN_PIPELINE_WAVEFRONT)
print(st["kernel_ms"], st["shade_launches"])
PY
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/serial/prof -o s -- python3 /tmp/serial_once.py > $ROOT/gpurun_out/serial/run.log 2>&1
cd $ROOT
find gpurun_out/serial/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/serial/kernel_stats.csv \;
find gpurun_out/serial/prof -name "*kernel_trace.csv" -delete
python tools/kernel_stats_summary.py gpurun_out/serial
