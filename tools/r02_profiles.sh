# End-of-round measurement set (1x MI355X): bench lines, rocprofv3 kernel stats (two streams / one stream), HBM-traffic
# counter passes.  Run from the repository root on the GPU box; results under gpurun_out/r2f_*.
R=$PWD
O=$R/gpurun_out
python bench.py --steps 30 --warmup 5 > $O/r2f_bench.json 2> $O/r2f_bench.err || exit 1
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --floatx float16 > $O/r2f_bench_f16.json 2> $O/r2f_bench_f16.err || exit 1
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --archi ssd_custom > $O/r2f_bench_custom.json 2>/dev/null || exit 1
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --archi up_sampling > $O/r2f_bench_ups.json 2>/dev/null || exit 1
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --archi ssd_custom --floatx float16 > $O/r2f_bench_custom_f16.json 2>/dev/null || exit 1
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --archi up_sampling --floatx float16 > $O/r2f_bench_ups_f16.json 2>/dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2f_prof2 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $O/r2f_prof2.json 2> $O/r2f_prof2.err || exit 1
DJ_SIDE_WGRAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2f_prof1 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $O/r2f_prof1.json 2> $O/r2f_prof1.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2f_prof16 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --floatx float16 > $O/r2f_prof16.json 2> $O/r2f_prof16.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r2f_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/r2f_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r2f_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/r2f_write.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r2f_fetch16 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --floatx float16 > /dev/null 2> $O/r2f_fetch16.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r2f_write16 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --floatx float16 > /dev/null 2> $O/r2f_write16.err || exit 1
cd $R
echo done
