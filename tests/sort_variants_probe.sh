#!/bin/bash
# GPU box: one radix pass of the k-mer sort at the configs[1] record count — the product kernel against its measurement variants
# (make sort_variants): the least a decoupled look-back would add per tile (digit row published + N predecessor rows read
# back, no waiting), and the payload staged as two u32 planes.   usage: tests/sort_variants_probe.sh OUT.txt
out=${1:-gpurun_out/sort_variants.txt}
: > $out
for v in sort_bench sort_bench_lb4 sort_bench_lb16 sort_bench_lb64 sort_bench_2plane; do
  [ -x tests/harness/bin/$v ] || continue
  echo "== $v" | tee -a $out
  tests/harness/bin/$v 448712444 28 2>&1 | tail -2 | tee -a $out
done
tests/harness/bin/sort_bench_lb16 5000000 28 | tail -1 | tee -a $out
tests/harness/bin/sort_bench_2plane 5000000 28 | tail -1 | tee -a $out
