#!/bin/bash
# round 5, call g: the drop-in engine at batch 1 (two transfers per call instead of a dozen): plug-in tests + host latency per call
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_contacts.py -q -m gpu -k "engine or plugin or rollout or world" 2>&1 | grep -E "passed|failed|^FAILED|Error" | cut -c1-300
timeout 300 python tools/experiments/engine_latency.py > $O/r05_g_engine_latency.json 2> $O/r05_g_engine_latency.err; echo "latency rc=$?"; python - <<'PY'
import json
j = json.load(open("gpurun_out/r05_g_engine_latency.json"))
for k, v in j["scenes"].items(): print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
PY
tail -2 $O/r05_g_engine_latency.err
