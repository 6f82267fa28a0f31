# same-binary A/B of the dense GEMM's tile order (UC_GEMM_GROUP_M: row panels per L2-sharing tile group) on the headline forward
for g in 1 2 4 8 16 32; do
  UC_GEMM_GROUP_M=$g python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-reference-policy --no-extra-legs > gpurun_out/r6_f_group_m_$g.json 2>/dev/null
done
