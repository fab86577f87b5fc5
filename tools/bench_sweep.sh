mkdir -p gpurun_out/r02
O=gpurun_out/r02/sweep.txt; : > $O
run(){ echo "### $*" >> $O; timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '^{"metric"' >> $O; }
run --steps 10 --warmup 3
run --steps 10 --warmup 3 --normalize posture
run --steps 10 --warmup 3 --encoding rgb8
run --steps 10 --warmup 3 --encoding rgb8 --normalize posture
run --steps 10 --warmup 3 --input bgra
run --steps 40 --warmup 5 --config C3
run --steps 40 --warmup 5 --config C2
run --steps 10 --warmup 3 --config C5
run --steps 20 --warmup 3 --batch 64
run --steps 20 --warmup 3 --batch 128
run --steps 6 --warmup 2 --batch 512
run --steps 40 --warmup 5 --stages segment
run --steps 6 --warmup 2 --input host-bgra
run --steps 6 --warmup 2 --input host-gray
run --steps 10 --warmup 3 --cnn-mode fp32
run --steps 10 --warmup 3 --cnn-mode bf16x6
run --steps 10 --warmup 3 --force-dist
run --steps 10 --warmup 3 --no-pipeline
cat $O
