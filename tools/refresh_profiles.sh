#!/bin/bash
# Everything a round commits under profiles/ from ONE build, in one call on the GPU box (run from the repo root under gpurun):
#   tools/refresh_profiles.sh r2_i
# kernel-trace summary + PMC passes of the forward (tools/profile_round.sh), the traffic record moved to where bench.py looks for
# it, the full default bench line, the kernel-trace summary of the training step and its bench line.  Outputs: gpurun_out/prof_<tag>/
# (the rocprofv3 databases are deleted once summarised: gpurun_out/ travels back and is size-limited).
set -u
tag=$1
export TMPDIR=/tmp
root=$(pwd)
out=gpurun_out/prof_$tag
bash tools/profile_round.sh $tag pmc > /dev/null 2>&1
cp $out/${tag}_pmc_traffic.json profiles/r6_pmc_traffic.json
rm -rf $out/trace $out/pmc_*
python bench.py --steps 20 --warmup 3 > $out/${tag}_bench_forward_pairs128.json 2> $out/bench_forward.err
cat $out/${tag}_bench_forward_pairs128.json
# (the trace runs single-stream: per-kernel durations are only defined without overlap; the bench line below is the default two-stream step)
cmd="python $root/bench.py --mode train --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --single-stream"
(cd /tmp && rocprofv3 --kernel-trace --stats -d $root/$out/train -o ${tag}_train -- $cmd) > $out/train.log 2>&1
db=$(ls $out/train/*/*_results.db $out/train/*_results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py $db > $out/${tag}_train_step_kernel_stats_dpt_pairs64.md
rm -rf $out/train
python bench.py --mode train --steps 10 --warmup 3 > $out/${tag}_bench_train_dpt_pairs64.json 2> $out/bench_train.err
cat $out/${tag}_bench_train_dpt_pairs64.json
# SQ counters of the bf16 attention forward (VERDICT r3 next #9 asks for the evidence): tools/pmc_attn.sh, eight-wave workgroups
bash tools/pmc_attn.sh "8" > $out/${tag}_probe_attention_sq_counters.txt 2>&1
rm -rf gpurun_out/pmc_attn
# the per-shape dense GEMM table and the two probes of the round
python tools/bench_model_gemms2.py 128 auto,2,6,7 enc,dec 0.4 > $out/${tag}_model_gemm_shapes.txt 2>&1
python tools/bench_attention_ab.py > $out/${tag}_attention_ab_same_box.txt 2>&1
python tools/bench_attention_bwd.py > $out/${tag}_attention_bwd.txt 2>&1
# round 6: kernel traces of the two other north-star forwards (224 x 224 at 256 pairs, DINOv2-518 at 32 pairs), single stream
base="--steps 3 --warmup 2 --single-stream --no-cpu-baseline --no-roofline --no-reference-policy --no-extra-legs"
for cfg in "224_pairs256:--img 224 --pairs 256" "dinov2_518_pairs32:--encoder dinov2 --pairs 32"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $root/$out/t_$name -o t_$name -- python $root/bench.py $flags $base) > $out/trace_$name.log 2>&1
  db=$(ls $out/t_$name/*/*_results.db $out/t_$name/*_results.db 2>/dev/null | head -1)
  python tools/rocpd_stats.py $db > $out/${tag}_forward_kernel_stats_$name.md
  python tools/rocpd_dispatches.py $db > $out/${tag}_forward_dispatches_$name.txt
  rm -rf $out/t_$name
done
python tools/bench_conv_flat.py > $out/${tag}_conv_flat_ab.txt 2>&1
if [ -x tools/_bin/dma_seg ]; then ./tools/_bin/dma_seg > $out/${tag}_probe_dma_seg.txt 2>&1; fi
if [ -x tools/_bin/mfma_valu ]; then ./tools/_bin/mfma_valu > $out/${tag}_probe_mfma_valu.txt 2>&1; fi
exit 0
