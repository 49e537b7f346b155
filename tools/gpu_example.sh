# run the offline example on a random-init 2-layer model (GPU)
python - <<'PY'
import sys; sys.path.insert(0, ".")
from oracle import synth
cfg = synth.make_config(**synth.SMALL128)
synth.write_model_dir("/tmp/swl_example_model", cfg, synth.make_state_dict(cfg, seed=0))
PY
python examples/offline.py --model-path /tmp/swl_example_model --steps 8
