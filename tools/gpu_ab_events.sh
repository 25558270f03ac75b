O=gpurun_out/r02_ab; mkdir -p $O
timeout 300 python tools/bench_configs.py --skip-1m --pairs 2 --guesses 2 2>/dev/null | grep resident_iteration | cut -c1-200 > $O/ev_on.log
MOLA_ICP_NO_EVENTS=1 timeout 300 python tools/bench_configs.py --skip-1m --pairs 2 --guesses 2 2>/dev/null | grep resident_iteration | cut -c1-200 > $O/ev_off.log
echo "events on"; cat $O/ev_on.log; echo "events off"; cat $O/ev_off.log
