set -eu
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --steps 30 --warmup 5 --schedule serial --no-cpu-baseline --no-hbm-probe --no-end-to-end --sustain-seconds 0 --first-encode-reps 0 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); k=d['kernel_ms_per_step']; print(' ', d['config']['workload'][:24], 'hbma ms', round(k['hbma'],4), 'frac', round(d['roofline']['frac'],3))"; }
cp scalable_video_codec_amd/libsvc_hip.so /tmp/asbuilt.so; trap "cp /tmp/asbuilt.so scalable_video_codec_amd/libsvc_hip.so" EXIT
for rep in 1 2; do
  cp scalable_video_codec_amd/_ab_hbma_l0c3_libsvc_hip.so scalable_video_codec_amd/libsvc_hip.so; echo "== hbma_l0c3: level 0's tracked rows from LDS, transfer free (timing only)"; run --config C3b-1080p-4L-dct8-quant; run --config C5-4k-4L-dct16
  cp /tmp/asbuilt.so scalable_video_codec_amd/libsvc_hip.so; echo "== as built"; run --config C3b-1080p-4L-dct8-quant; run --config C5-4k-4L-dct16
done
