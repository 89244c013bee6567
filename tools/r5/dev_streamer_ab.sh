#!/bin/bash
# Runs ON THE GPU BOX: the headline with the key frames' rows converted by the row streamer (SCPR_DEV_STREAMER=1) against k_unpack32 afterwards
for p in 0 1 0 1; do
  SCPR_DEV_STREAMER=$p python3 bench.py --no-others --steps 5 --warmup 1 --no-cpu --no-host-boundary 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('streamer $p', d['value'],d['ms_per_step'],d['config']['dec_MPix_s_rank0'],{k:v for k,v in d['config']['stage_ms_per_step'].items() if k in ('decode','unpack')}, d['parity']['lossless_roundtrip'])"
done
