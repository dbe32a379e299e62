for o in "" "--opt bwd_mode=1" "--opt defer_colour=0" "--opt bwd_mode=1 --opt defer_colour=0" "--opt bwd_mode=2" "--opt own_sort=0" "--opt fused_tree=0"; do
  python bench.py --workload s10k --no-cpu-baseline --no-both-paths --no-vary --min-seconds 1 --steps 200 --check-sum $o 2>/dev/null | python tools/benchline.py "s10k [$o]"
done
