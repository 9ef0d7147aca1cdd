# Round evidence (GPU box): the device walks of BASELINE configs[1] for the seeds bench.py uses on 1..8 GPUs (2..9): speculative,
# exact and host-tail modes must agree; the fingerprints are those of the host walk (profiles/r02_walk_check_seeds.log,
# bench.py KNOWN_PATHS).
mkdir -p gpurun_out/evidence
: > gpurun_out/evidence/walk_check_seeds.log
for s in 2 3 4 5 6 7 8 9; do
  echo "seed $s" >> gpurun_out/evidence/walk_check_seeds.log
  python tests/walk_check.py --reads 100000 --ref-len 50000000 --seed $s --no-host --modes speculative,exact,hosttail 2>/dev/null | grep -v "^\[" >> gpurun_out/evidence/walk_check_seeds.log
done
grep -c "ALL EQUAL" gpurun_out/evidence/walk_check_seeds.log
