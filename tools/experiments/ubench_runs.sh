B=tools/experiments/ubench_runs.bin
echo "--- 64-byte aligned starts (mis=2) vs line-aligned (0) vs random (1)"
for run in 16 32 48 64; do for mis in 0 2 1; do $B $run 256 $mis 65536 0 0 | tail -1; done; done
echo "--- with the read stream"
for run in 16 32 64; do for mis in 0 2 1; do $B $run 256 $mis 65536 0 1 | tail -1; done; done
