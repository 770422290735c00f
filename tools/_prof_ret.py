import cProfile, pstats, io, os, sys, tempfile
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import numpy as np, torch
from contextlib import redirect_stdout
import train_uci13_demo as demo
root = tempfile.mkdtemp(prefix="r4d_ret_")
base, ret = demo.build_workdir(root); os.chdir(root)
import main_retriever
out = os.path.join(root, "out")
argv = (f"--dataset UCI_13 --timestamp 12 --output_dir {out} --model_type gpt2 --model_name_or_path gpt2 "
        f"--train_data_file {base}/train.link_prediction --eval_data_file {base}/val.link_prediction "
        f"--eval_data_gt_file {ret}/val_score.retrieval --test_data_file {base}/test.link_prediction "
        f"--test_data_gt_file {ret}/test_score.retrieval --n_layer 4 --n_head 2 --n_embed 512 --block_size 512 --seed 42 --topK 5 "
        f"--train_pair_data_file {ret}/train_index.retrieval --num_train_epochs 1 --per_gpu_train_batch_size 64 --do_train --patience 50").split()
with redirect_stdout(io.StringIO()):
    main_retriever.main(argv)
ev = [a for a in argv if a != "--do_train"] + ["--do_eval", "--eval_all_checkpoints"]
with redirect_stdout(io.StringIO()):
    main_retriever.main(ev)
pr = cProfile.Profile(); pr.enable()
with redirect_stdout(io.StringIO()):
    main_retriever.main(ev)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40); print(s.getvalue()[:9000])
