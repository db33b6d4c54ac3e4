# ON THE GPU BOX: kernel time of the Winograd products per layer class for the ablation builds of wino_mm_kernel (variants/libfte_abl<n>.so)
for l in base abl1 abl2 abl4 abl3 abl7; do
  if [ "$l" = base ]; then unset FTE_LIB; else export FTE_LIB=variants/libfte_$l.so; fi
  echo "== $l"; python scripts/dev/wino_bench.py 512 10 1,2,3 2>/dev/null | grep "winograd" | grep -v wgrad | sed 's/direct-equivalent//' | cut -c1-140
done
