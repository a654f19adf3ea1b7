# Bench lines of BASELINE configs 1-5 on one GPU (config 2 with the CPU baseline = the default bench line): gpurun -- 'TAG=r02b bash tools/bench_configs.sh'
cd $GRAFT_REPO_ROOT
TAG=${TAG:-r02b}
mkdir -p gpurun_out
timeout 300 python bench.py --config 1 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_config1.json 2> gpurun_out/${TAG}_bench_config1.err
timeout 600 python bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_config3.json 2> gpurun_out/${TAG}_bench_config3.err
timeout 900 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_config4.json 2> gpurun_out/${TAG}_bench_config4.err
timeout 600 python bench.py > gpurun_out/${TAG}_bench_config2.json 2> gpurun_out/${TAG}_bench_config2.err
timeout 1200 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_config5.json 2> gpurun_out/${TAG}_bench_config5.err
for c in 1 2 3 4 5; do cut -c1-700 gpurun_out/${TAG}_bench_config$c.json; tail -2 gpurun_out/${TAG}_bench_config$c.err; done
# configs 3 / 4 on the atrium at SURVEY 8(d)'s ~262 k triangles as well (round 6): SIZE262=1
if [ -n "$SIZE262" ]; then
timeout 700 python bench.py --config 3 --atrium-segments 112 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_config3_262k.json 2> gpurun_out/${TAG}_bench_config3_262k.err
timeout 1000 python bench.py --config 4 --atrium-segments 112 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_config4_262k.json 2> gpurun_out/${TAG}_bench_config4_262k.err
for c in 3 4; do cut -c1-700 gpurun_out/${TAG}_bench_config${c}_262k.json; tail -2 gpurun_out/${TAG}_bench_config${c}_262k.err; done
fi
