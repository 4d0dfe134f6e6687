for a in $ABL; do
  timeout -k 10 120 python bench.py --no-cpu-baseline --steps 2 --ablate $a > gpurun_out/abl_$a.json 2> gpurun_out/abl_$a.err || echo "fail $a"
done
