"""Training driver with the flags, log lines, spike rollback and checkpoint cadence of the reference train.py
(:43-300), driving the MI355X engine instead of a TF session.

    python train.py --hparams="outputs_per_step=5" --train_data=THCHS --GPUs_id=[0]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...   (data parallel, RCCL)

Reads ./train_npy_data_dict.json {name: metadata_path}; the speaker count is parsed from `_id_num_<n>` in the
path (reference :76-85).  Out of scope here: TF summaries, Griffin-Lim audio and
alignment dumps at checkpoints.  The reference's tower loop over --GPUs_id never trained more than one tower
(SURVEY.md fact 5); multi-GPU here = one process per GPU with gradient all-reduce.
"""
import argparse
import ast
import json
import math
import os
import re
import time
import traceback

import torch

from datasets.datafeeder_npy import DataFeeder as DataFeeder_npy
from hparams import hparams, hparams_debug_string
from models import create_model
from models.tacotron import GlobalStep
from util import ValueWindow, infolog
from util.coordinator import Coordinator

log = infolog.log


def train(log_dir, args):
    checkpoint_path = os.path.join(log_dir, 'model.ckpt')
    log('Checkpoint path: %s' % checkpoint_path)
    log('Using model: %s' % args.model)
    log(hparams_debug_string())

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    GPUs_id = ast.literal_eval(args.GPUs_id)
    hparams.num_GPU = max(len(GPUs_id), world)
    local = int(os.environ.get('LOCAL_RANK', str(GPUs_id[0])))
    torch.cuda.set_device(local)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))

    coord = Coordinator()
    if args.data_type != 'npy':
        raise TypeError('only --data_type=npy is supported (the TFRecord feeder is TF-queue specific)')
    with open('./train_npy_data_dict.json', 'r') as f:
        train_data_dict = json.load(f)
    file_list, id_num = [], 0
    for item in args.train_data.split(','):
        file_list.append(train_data_dict[item])
        id_num += int(re.findall('[.]*\\_id\\_num\\_([0-9]+)[.]+', train_data_dict[item])[0])
    log('train data:%s' % args.train_data)
    feeder = DataFeeder_npy(hparams, file_list, coord)

    global_step = GlobalStep()
    model = create_model(args.model, hparams)
    model.initialize(inputs=feeder.inputs, input_lengths=feeder.input_lengths, mel_targets=feeder.mel_targets,
                     linear_targets=feeder.linear_targets, identities=feeder.identities, id_num=id_num,
                     device='cuda:%d' % local)
    model.add_loss()
    model.add_optimizer(global_step)
    model.engine.world = world

    step = 0
    steps_run = 0                    # --max_steps also bounds the number of iterations, so a run that keeps rolling back ends
    time_window = ValueWindow(250)
    loss_window = ValueWindow(1000)
    try:
        if args.restore_step:
            restore_path = '%s-%d' % (checkpoint_path, args.restore_step)
            model.load_state_dict(torch.load(restore_path, weights_only=True))
            log('Resuming from checkpoint: %s' % restore_path)
        else:
            log('Starting new training run')
        feeder.start_in_session(None)
        while not coord.should_stop():
            start_time = time.time()
            out = model.run_step()
            if out is None:
                break
            step, loss, _, loss_regularity = out
            time_window.append(time.time() - start_time)
            loss_window.append(loss)
            log('Step %-7d [%.03f avg_sec/step,  loss=%.05f,  avg_loss=%.05f,  lossw=%.05f]' % (
                step, time_window.average, loss, loss_window.average, loss_regularity))
            # if the gradient seems to explode, then restore to the previous step (reference :154-160).  Under data
            # parallelism the replicas must take the same branch: the decision is the OR over ranks (each rank only sees the
            # loss of its own shard), and a rank reads a checkpoint only after rank 0 has finished writing it.
            spike = bool(loss > 2 * loss_window.average or math.isnan(loss))
            if world > 1:
                flag = torch.tensor([1 if spike else 0], device='cuda:%d' % local, dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                spike = bool(flag.item())
            if spike:
                log('recover to the previous checkpoint')
                restore_step = int((step - 10) / args.checkpoint_interval) * args.checkpoint_interval
                restore_path = '%s-%d' % (checkpoint_path, restore_step)
                # a missing checkpoint raises (like saver.restore at reference :159) instead of training on with NaN weights
                model.load_state_dict(torch.load(restore_path, weights_only=True))
                steps_run += 1
                if args.max_steps and steps_run >= args.max_steps:
                    coord.request_stop()
                continue
            if step % args.checkpoint_interval == 0:
                if rank == 0:
                    log('Saving checkpoint to: %s-%d' % (checkpoint_path, step))
                    torch.save(model.state_dict(), '%s-%d' % (checkpoint_path, step))
                if world > 1:
                    dist.barrier()          # the file is complete before any rank may roll back to it
            steps_run += 1
            if args.max_steps and (step >= args.max_steps or steps_run >= args.max_steps):
                coord.request_stop()
    except Exception as e:
        log('Exiting due to exception: %s' % e)
        traceback.print_exc()
        coord.request_stop(e)


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument('--base_dir', default='./logs/')
    parser.add_argument('--model', default='tacotron')
    parser.add_argument('--summary_interval', type=int, default=100, help='Steps between running summary ops.')
    parser.add_argument('--tf_log_level', type=int, default=1, help='(ignored) Tensorflow C++ log level.')
    parser.add_argument('--hparams', default='', help='Hyperparameter overrides as a comma-separated list of name=value pairs')
    parser.add_argument('--restore_step', type=int, help='Global step to restore from checkpoint.')
    parser.add_argument('--checkpoint_interval', type=int, default=1000, help='Steps between writing checkpoints.')
    parser.add_argument('--GPUs_id', default='[0]', help='The GPUs\' id list that will be used. Default is 0')
    parser.add_argument('--description', default=None, help='description of the model')
    parser.add_argument('--train_data', type=str, default='THCHS', help='training datas to be used, comma-separated')
    parser.add_argument('--data_type', type=str, default='npy', help='tfrecord or npy')
    parser.add_argument('--max_steps', type=int, default=0, help='(extension) stop after this many steps')
    args = parser.parse_args()
    log_dir = os.path.join(args.base_dir, 'logs-%s-%s' % (args.model, args.description))
    os.makedirs(log_dir, exist_ok=True)
    infolog.init(os.path.join(log_dir, 'train.log'))
    hparams.parse(args.hparams)
    train(log_dir, args)


if __name__ == '__main__':
    main()
