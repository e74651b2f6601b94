for T in 28 32 36; do
RT_REFILL_MIN=$T PYTHONPATH=. timeout -k 10 100 python exp/fine_sweep.py 2>&1 | grep "full" | sed "s/^/T=$T /" || exit 1
done
for T in 28 32 36; do
RT_REFILL_MIN=$T PYTHONPATH=. timeout -k 10 100 python exp/fine_sweep.py 2>&1 | grep "full" | sed "s/^/T=$T /" || exit 1
done
