# One-off measurement (GPU box): a cold bin/pagraph process on the text of BASELINE configs[1], with the library's timing lines.
export LC_ALL=C
O=gpurun_out/r04m; mkdir -p $O
python tests/c2_text_runs.py $O/text_run.json --skip-reference --keep-text > $O/gen.log 2>&1
D=/dev/shm/c2_text
for i in 1 2 3 4; do
  out=/dev/shm/c2_out_$i; rm -rf $out; mkdir -p $out
  t0=$(date +%s%N)
  env PAGRAPH_TIMING=1 $EXTRA aligngraph2_amd/bin/pagraph -t 16 -r dummy -k $D/kmer.bin -c $D/ctg.fasta -R $D/ref.fasta -p $D -a $D/aln -o $out -r 50 --epsilon 10 -v 2 > $O/run_$i.out 2> $O/run_$i.err
  echo "run $i wall $(( ($(date +%s%N) - t0) / 1000000 )) ms"
  sleep 6
done
grep "timing\] \(load\|device memory\|prepare\|graph build\|successor records [0-9]\|walks\|last block\|  \)" $O/run_4.err | cut -c1-200
rm -rf /dev/shm/c2_text /dev/shm/c2_out_1 /dev/shm/c2_out_2 /dev/shm/c2_out_3 /dev/shm/c2_out_4
