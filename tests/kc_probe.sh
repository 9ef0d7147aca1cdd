# One-off measurement (GPU box): the device k-mer counter at configs[1] with the table filled in 1, 2, 4, 8, 16 slices of the code range.
for sl in 1 2 4 8 16 default; do
  if [ $sl = default ]; then unset PAG_KC_SLICES; else export PAG_KC_SLICES=$sl; fi; python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-file-to-file 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slices $sl', d['config']['kmer_counter_on_device'])"
done
