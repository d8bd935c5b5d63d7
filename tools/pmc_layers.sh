# Per-layer HBM traffic of the conv family (1x MI355X): two counter passes + the ordered call list.  $1 = float32 | float16
R=$PWD; O=$R/gpurun_out; FX=${1:-float32}
PROFILE_LAYERS_JSON=$O/pl_calls_$FX.json python tools/profile_layers.py deconv 32 $FX > $O/pl_layers_$FX.txt 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pl_fetch_$FX -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --floatx $FX > /dev/null 2> $O/pl_fetch_$FX.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pl_write_$FX -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --floatx $FX > /dev/null 2> $O/pl_write_$FX.err || exit 1
cd $R
python tools/pmc_layers.py $(find $O/pl_fetch_$FX -name '*counter_collection.csv' | head -1) $(find $O/pl_write_$FX -name '*counter_collection.csv' | head -1) $O/pl_calls_$FX.json > $O/pl_traffic_$FX.txt 2>&1
head -40 $O/pl_traffic_$FX.txt
