set -x
mkdir -p gpurun_out/final4
python -m pytest tests -q -m gpu 2>&1 | tail -3 > gpurun_out/final4/pytest_gpu.txt
python bench.py > gpurun_out/final4/bench_c2.json 2> gpurun_out/final4/bench_c2.err
python bench.py --workload atrium_1920x1080_64spp_md8 > gpurun_out/final4/bench_c3.json 2> gpurun_out/final4/bench_c3.err
python bench.py --workload glassroom_1920x1080_512spp_md16 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/final4/bench_c4.json 2> gpurun_out/final4/bench_c4.err
R=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final4 -o c2 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/final4/prof_c2.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final4 -o c3 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload atrium_1920x1080_64spp_md8 > $R/gpurun_out/final4/prof_c3.log 2>&1
cd $R
PHIP_POOL=8388608 python tools/pmc_traffic.py cornell_1024x1024_256spp gpurun_out/final4/traffic_c2.json 64 > gpurun_out/final4/traffic_c2.log 2>&1
python tools/pmc_traffic.py atrium_1920x1080_64spp_md8 gpurun_out/final4/traffic_c3.json 16 > gpurun_out/final4/traffic_c3.log 2>&1
PHIP_POOL=8388608 python tools/pmc_traffic.py glassroom_1920x1080_512spp_md16 gpurun_out/final4/traffic_c4.json 32 > gpurun_out/final4/traffic_c4.log 2>&1
python __graft_entry__.py smoke > gpurun_out/final4/smoke.txt 2>&1
ls gpurun_out/final4
