# One-off measurement (GPU box): BASELINE configs[1] as text through bin/pagraph with the ALN column classes from the host
# parser and from the device (PAGRAPH_DEVICE_INGEST=1, k_ingest.hip); outputs must be the same bytes.
set -e
export LC_ALL=C
O=gpurun_out/r04k; mkdir -p $O
python tests/c2_text_runs.py $O/text_run.json --skip-reference --keep-text > $O/gen.log 2>&1
D=/dev/shm/c2_text
for mode in host device host device; do
  out=/dev/shm/c2_out_$mode; rm -rf $out; mkdir -p $out
  if [ $mode = device ]; then export PAGRAPH_DEVICE_INGEST=1; else unset PAGRAPH_DEVICE_INGEST; fi
  t0=$(date +%s%N)
  env PAGRAPH_TIMING=1 aligngraph2_amd/bin/pagraph -t 16 -r dummy -k $D/kmer.bin -c $D/ctg.fasta -R $D/ref.fasta -p $D -a $D/aln -o $out -r 50 --epsilon 10 -v 2 > $O/run_$mode.out 2> $O/run_$mode.err
  echo "$mode wall $(( ($(date +%s%N) - t0) / 1000000 )) ms"
  grep "ALN \|load block\|load global\|reserved" $O/run_$mode.err | tail -6
done
(cd /dev/shm/c2_out_host && sha256sum * | sort) > $O/host.sha; (cd /dev/shm/c2_out_device && sha256sum * | sort) > $O/device.sha
if cmp -s $O/host.sha $O/device.sha; then echo "outputs identical: $(wc -l < $O/host.sha) files"; else echo "OUTPUTS DIFFER"; fi
rm -rf /dev/shm/c2_text /dev/shm/c2_out_host /dev/shm/c2_out_device
