python -m pytest tests -m gpu -x -q 2>&1 | tail -5
# bench through the distributed launcher with one rank (exercises init_process_group + barrier path)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -3 | cut -c1-400
