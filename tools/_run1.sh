mkdir -p gpurun_out
export TMPDIR=/tmp
(time python -m pytest tests -m gpu -x -q) > gpurun_out/r02_pytest1.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_driver.json 2> gpurun_out/r02_bench_driver.err
python bench.py --gpus 1 --steps 20 --warmup 5 --precondition-s 0 --no-cpu-baseline > gpurun_out/r02_bench_noprecond.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --config cfg4 > gpurun_out/r02_bench_cfg4.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --config cfg3nc > gpurun_out/r02_bench_cfg3nc.json 2>/dev/null
python tests/tools/ref_rounding_stats.py > gpurun_out/r02_ref_rounding_stats.txt 2>&1
tail -5 gpurun_out/r02_pytest1.log
