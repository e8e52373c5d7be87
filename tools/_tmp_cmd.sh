bash tools/trace_minkunet.sh 1000000 8 | tail -42
bash tools/trace_minkunet.sh 200000 8 | tail -42
