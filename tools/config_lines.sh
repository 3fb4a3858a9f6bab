set -e
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench_line.err; tail -1 gpurun_out/r03_bench_line.err
: > gpurun_out/r03_config_lines.jsonl
python bench.py --stage 1 --no-cpu-baseline >> gpurun_out/r03_config_lines.jsonl 2> gpurun_out/l1.err; tail -1 gpurun_out/l1.err
python bench.py --stage 1 --pairs-per-gpu 1024 --no-cpu-baseline >> gpurun_out/r03_config_lines.jsonl 2> gpurun_out/l2.err; tail -1 gpurun_out/l2.err
python bench.py --vision-model openai/clip-vit-large-patch14 --text-model gpt2-large --seq-len 256 --pairs-per-gpu 64 --no-cpu-baseline --steps 5 --warmup 2 >> gpurun_out/r03_config_lines.jsonl 2> gpurun_out/l3.err; tail -1 gpurun_out/l3.err
python bench.py --vision-model openai/clip-vit-large-patch14 --text-model gpt2-xl --seq-len 256 --pairs-per-gpu 32 --no-cpu-baseline --steps 5 --warmup 2 >> gpurun_out/r03_config_lines.jsonl 2> gpurun_out/l4.err; tail -1 gpurun_out/l4.err
python bench.py --stage 1 --vision-model openai/clip-vit-large-patch14 --text-model gpt2-xl --seq-len 256 --pairs-per-gpu 128 --no-cpu-baseline --steps 5 --warmup 2 >> gpurun_out/r03_config_lines.jsonl 2> gpurun_out/l5.err; tail -1 gpurun_out/l5.err
python bench.py --padded --pairs-per-gpu 256 --no-cpu-baseline >> gpurun_out/r03_config_lines.jsonl 2> gpurun_out/l6.err; tail -1 gpurun_out/l6.err
