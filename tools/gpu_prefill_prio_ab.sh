mkdir -p gpurun_out; : > gpurun_out/prefill_prio_ab.jsonl
for round in 1 2 3; do
  for shape in c3 c4; do
    for v in cur noprio invprio; do
      lib=$PWD/swiftllm_amd/csrc/libswiftllm_hip.so; [ $v != cur ] && lib=$PWD/swiftllm_amd/csrc/libswiftllm_hip_$v.so
      SWIFTLLM_HIP_LIB=$lib timeout 300 python tools/prefill_attn_micro.py --shape $shape --iters 20 2>/dev/null | sed "s/^{/{\"variant\": \"$v\", \"round\": $round, /" >> gpurun_out/prefill_prio_ab.jsonl
    done
  done
done
python - <<'P'
import json, collections
acc = collections.defaultdict(list)
for l in open("gpurun_out/prefill_prio_ab.jsonl"):
    d = json.loads(l); acc[(d["shape"], d["variant"])].append(d["TFLOPs"])
for k in sorted(acc): print(k, acc[k])
P
