# End-of-round measurement set (1x MI355X): bench lines, rocprofv3 kernel stats (two streams / one stream, fp32 and
# float16), HBM-traffic counter passes with a calibration of the counters for both access widths.  Run from the repository
# root on the GPU box; results under gpurun_out/r3z_*.
R=$PWD
O=$R/gpurun_out
python bench.py --steps 30 --warmup 5 > $O/r3z_bench.json 2> $O/r3z_bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3z_prof2 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-secondary > $O/r3z_prof2.json 2> $O/r3z_prof2.err || exit 1
DJ_SIDE_WGRAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3z_prof1 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-secondary > $O/r3z_prof1.json 2> $O/r3z_prof1.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3z_prof_mfma -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-secondary --floatx float32_mfma > $O/r3z_prof_mfma.json 2> $O/r3z_prof_mfma.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3z_prof_x3 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-secondary --floatx float32x3 > $O/r3z_prof_x3.json 2> $O/r3z_prof_x3.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3z_prof16 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-secondary --floatx float16 > $O/r3z_prof16.json 2> $O/r3z_prof16.err || exit 1
DJ_SIDE_WGRAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3z_prof16_1 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-secondary --floatx float16 > $O/r3z_prof16_1.json 2> $O/r3z_prof16_1.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r3z_cal_fetch -- python $R/tools/pmc_calibrate.py > /dev/null 2> $O/r3z_cal_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r3z_cal_write -- python $R/tools/pmc_calibrate.py > /dev/null 2> $O/r3z_cal_write.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r3z_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $O/r3z_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r3z_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $O/r3z_write.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r3z_fetch16 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --floatx float16 > /dev/null 2> $O/r3z_fetch16.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r3z_write16 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --floatx float16 > /dev/null 2> $O/r3z_write16.err || exit 1
cd $R
CAL=$(python tools/pmc_calib_factors.py $(find $O/r3z_cal_fetch -name '*counter_collection.csv' | head -1) $(find $O/r3z_cal_write -name '*counter_collection.csv' | head -1))
echo "calibration (known / reported bytes): fp32 fetch, fp32 write, fp16 fetch, fp16 write = $CAL" | tee $O/r3z_calibration.txt
set -- $CAL
python tools/pmc_traffic.py $(find $O/r3z_fetch -name '*counter_collection.csv' | head -1) $(find $O/r3z_write -name '*counter_collection.csv' | head -1) $O/r3z_igemm_traffic.json $1 $2 > $O/r3z_traffic.txt
python tools/pmc_traffic.py $(find $O/r3z_fetch16 -name '*counter_collection.csv' | head -1) $(find $O/r3z_write16 -name '*counter_collection.csv' | head -1) $O/r3z_igemm_traffic_f16.json $3 $4 > $O/r3z_traffic16.txt
python tools/profile_layers.py deconv 32 float16 > $O/r3z_layers_f16.txt 2>&1
python tools/profile_layers.py deconv 32 float32 > $O/r3z_layers_f32.txt 2>&1
python tools/profile_layers.py deconv 32 float32_mfma > $O/r3z_layers_mfma.txt 2>&1
python tools/profile_layers.py deconv 32 float32x3 > $O/r3z_layers_x3.txt 2>&1
echo done
