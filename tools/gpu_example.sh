# run the offline example on a random-init 2-layer model (GPU)
python - <<'PY'
import sys; sys.path.insert(0, ".")
from oracle import synth
cfg = synth.make_config(**synth.SMALL128)
synth.write_model_dir("/tmp/swl_example_model", cfg, synth.make_state_dict(cfg, seed=0))
PY
python examples/offline.py --model-path /tmp/swl_example_model --steps 8
# ... the online example (Engine + scheduler, one streamed request) on the same model
python examples/online.py --model-path /tmp/swl_example_model --token-ids --output-len 8 --piggyback
# ... and the HTTP server through the `swiftllm` alias, one request with curl-equivalent python
python -m swiftllm.server.api_server --model-path /tmp/swl_example_model --port 8123 --max-batch-size 8 \
    --max-tokens-in-batch 512 --max-seqs-in-block-table 16 --max-blocks-per-seq 64 --num-cpu-blocks 8 \
    --gpu-mem-utilization 0.5 > /tmp/api_server.log 2>&1 &
SERVER=$!
python - <<'PY'
import json, time, urllib.request
for _ in range(120):
    try:
        req = urllib.request.Request("http://127.0.0.1:8123/generate", method="POST",
                                     data=json.dumps({"prompt_token_ids": [1, 2, 3, 4], "output_len": 5}).encode(),
                                     headers={"Content-Type": "application/json"})
        print("POST /generate ->", urllib.request.urlopen(req, timeout=30).read().decode())
        break
    except Exception as e:      # server still starting
        time.sleep(1)
else:
    print("api server did not answer"); print(open("/tmp/api_server.log").read()[-2000:])
PY
kill $SERVER
