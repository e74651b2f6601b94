for T in 8 16 24 32 40; do
RT_REFILL_MIN=$T timeout -k 10 120 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 > gpurun_out/b_rf$T.json 2> gpurun_out/b_rf$T.err || { echo "bench T=$T failed"; tail -3 gpurun_out/b_rf$T.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/b_rf$T.json'));print($T, d['ms_per_step'],d['roofline']['stage_ms_per_frame'])"
done
