mkdir -p gpurun_out/r06
B="python bench.py --steps 3 --warmup 2 --end-to-end --no-cpu-baseline --sustain-seconds 0 --first-encode-reps 0"
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d["end_to_end"]; print("bench end_to_end: passes", e["stream_encoder_fps"], e["stream_encoder_phases"]["d2h_GBps"], "GB/s | one stream", e["stream_encoder_long_stream"]["frames_per_s"], e["stream_encoder_long_stream"]["d2h_GBps"], "GB/s", "by pass", e["stream_encoder_phases"].get("d2h_GBps_by_pass"))'
$B 2>/dev/null | python -c "$pick"
timeout -k 10 300 python tools/probe_stream_d2h.py --reps 1 2>/dev/null | grep stream_main
$B 2>/dev/null | python -c "$pick"
timeout -k 10 300 python tools/probe_stream_d2h.py --reps 1 2>/dev/null | grep stream_main
