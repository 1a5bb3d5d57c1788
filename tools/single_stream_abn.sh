run() { SPEECHPLAYER_LIVE_TRACE=1 python tools/single_stream_probe.py 30 2>&1 | grep -o "kernel ([0-9.]* ms)\|pcm_equal\": [a-z]*\|ms_per_pull_median\": [0-9.]*" | tr -d '(' > /tmp/k0.txt; grep "pcm_equal\|median" /tmp/k0.txt | tr '\n' ' '
  grep kernel /tmp/k0.txt | awk '{print $2}' | tail -30 > /tmp/k.txt
  sort -n /tmp/k.txt | awk '{a[NR]=$1; s+=$1} END {printf "kernel: n=%d min %.2f p25 %.2f median %.2f p75 %.2f max %.2f mean %.3f\n", NR, a[1], a[int(NR/4)+1], a[int(NR/2)+1], a[int(3*NR/4)+1], a[NR], s/NR}'; }
for rep in 1 2; do
echo "== shipped"; run
for v in "$@"; do echo "== $v"; SPEECHPLAYER_LIB=nvspeechplayer_amd/lib/variants/libspeechPlayer_$v.so run; done
done
