export S5FXP_CGATE_FT32=1
python - <<PY
import numpy as np, torch, sys
sys.path.insert(0,'.')
from tests import test_gpu_parity as T
PY
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_forward_matches_oracle or any_sequence_length or grouped or live_states or decoder_carries" 2>&1 | tail -3
for w in 512 1024; do export S5FXP_WGS_CGATE=$w; echo "== FT32 WGS_CGATE=$w (x2 for the 32-frame grid)"; BENCH_ARGS="--steps 48 --no-one-batch-pass" bash tools/run_variants.sh base 2>&1 | grep -E "^==|cgate_p"; done
unset S5FXP_CGATE_FT32 S5FXP_WGS_CGATE
echo "== FT64"; BENCH_ARGS="--steps 48 --no-one-batch-pass" bash tools/run_variants.sh base 2>&1 | grep -E "^==|cgate_p"
