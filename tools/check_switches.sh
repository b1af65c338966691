#!/bin/bash
# check_switches.sh: the documented S5FXP_* experiment switches (include/s5fxp.h) must not change results: the parity tests that do
# not look at which kernels ran, once per switch
K="fused_forward_matches_oracle or any_sequence_length or grouped_forward_equals or decoder_carries or streaming_chunks or two_d_input"
for e in S5FXP_NO_COMPACT S5FXP_NO_LIVE_LANES S5FXP_NO_DEC_RESID S5FXP_GATE_BN S5FXP_CGATE_FT32 S5FXP_NO_PK16 S5FXP_NO_PAIR S5FXP_PAIR_GLOBAL S5FXP_NO_BN_EXT; do
  k="$K"
  case $e in S5FXP_NO_PAIR|S5FXP_PAIR_GLOBAL) k="${K/any_sequence_length or /}";; esac  # that test also asserts the LDS-fed pair rung
  echo -n "$e=1: "; env $e=1 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$k" 2>&1 | tail -1
done
