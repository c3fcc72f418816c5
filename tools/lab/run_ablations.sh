bin/synth_bam write /tmp/s50.bam 50000000 --threads 64 > /dev/null 2>&1
bin/synth_bam write /tmp/r50.bam 50000000 --threads 64 --realistic > /dev/null 2>&1
for f in /tmp/s50.bam /tmp/r50.bam; do
  echo "== $f"
  for v in base noload hotload nostore decode_only batch64 batch64_noload batch64_hotload; do
    REGTOOLS_AMD_INFLATE=lane timeout 60 tools/lab/bin/inflate_lab_$v $f 3 2>&1 | cut -c1-90
  done
done
