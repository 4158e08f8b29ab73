for w in 0 2 4 8; do for f in -1 0 1; do
  opts=""; [ $w -gt 0 ] && opts="tiles_wpb=$w"; [ $f -ge 0 ] && opts="$opts phase_flat=$f"
  echo "== $opts"; python tools/learned_phase_bench.py 4096 8192 65536 TT-ProDMP $opts 2>/dev/null | grep "| trajectory |"
done; done
