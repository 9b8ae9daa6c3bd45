set -x
O=gpurun_out/final7
mkdir -p $O
python -m pytest tests -q -m gpu 2>&1 | tail -3 > $O/pytest_gpu.txt
python -m pytest tests/test_ref_pin.py tests/test_golden.py -q 2>&1 | tail -3 > $O/pytest_ref_pin.txt
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
python bench.py --workload atrium_1920x1080_64spp_md8 > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --workload glassroom_1920x1080_512spp_md16 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
python bench.py --workload cornell_1024x1024_256spp_direct --no-cpu-baseline > $O/bench_c2_direct.json 2> $O/bench_c2_direct.err
python bench.py --workload atrium_1920x1080_64spp_direct4 --no-cpu-baseline > $O/bench_c3_direct4.json 2> $O/bench_c3_direct4.err
R=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/$O -o c2 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/prof_c2.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$O -o c3 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload atrium_1920x1080_64spp_md8 > $R/$O/prof_c3.log 2>&1
cd $R
PHIP_POOL=8388608 python tools/pmc_traffic.py cornell_1024x1024_256spp $O/traffic_c2.json 64 > $O/traffic_c2.log 2>&1
python tools/pmc_traffic.py atrium_1920x1080_64spp_md8 $O/traffic_c3.json 16 > $O/traffic_c3.log 2>&1
PHIP_POOL=8388608 python tools/pmc_traffic.py glassroom_1920x1080_512spp_md16 $O/traffic_c4.json 32 > $O/traffic_c4.log 2>&1
python __graft_entry__.py smoke > $O/smoke.txt 2>&1
ls $O
