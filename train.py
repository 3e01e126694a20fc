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
    torch.set_num_threads(4)         # no torch CPU compute on the path: keep the intra-op pool (sized by the HOST's CPU count) out of the way
    import torch.distributed as dist
    if world > 1:
        backend = os.environ.get('TACO_DIST_BACKEND', 'nccl')      # gloo: two ranks on ONE GPU (scripts/dp_two_ranks.py --train)
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

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

    # a tiny host-side (gloo) group carries the per-step agreement of the replicas: it must not sit on the GPU streams, where it
    # would force a device synchronisation per step
    ctl = dist.new_group(backend='gloo') if world > 1 else None

    def agree(*flags):
        """MAX over ranks of a few small integers (1 rank: identity).  Every rank takes the same branch afterwards."""
        if world == 1:
            return list(flags)
        t = torch.tensor(flags, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=ctl)
        return [int(x) for x in t]

    step = 0
    steps_run = 0                    # --max_steps also bounds the number of iterations, so a run that keeps rolling back ends
    time_window = ValueWindow(250)
    loss_window = ValueWindow(1000)
    saved = []                       # checkpoints on disk, oldest first (max_to_keep=5, keep_checkpoint_every_n_hours=8: reference :117)
    kept_forever_at = time.time()
    try:
        if args.restore_step:
            restore_path = '%s-%d' % (checkpoint_path, args.restore_step)
            model.load_state_dict(torch.load(restore_path, weights_only=True))
            log('Resuming from checkpoint: %s' % restore_path)
        else:
            log('Starting new training run')
        feeder.start_in_session(None)
        # The loop of reference train.py:139-152, pipelined one step deep: step k+1 is submitted BEFORE the host waits for the
        # scalars of step k (one 128-byte copy behind an event), so the GPU never idles while the host enqueues.  The host knows
        # the number every submitted step will get (global_step increments once per step), so the checkpoint of step k is cloned
        # on the device right behind step k although it is written to disk only after step k's loss has been looked at; the
        # rollback target int((step-10)/interval)*interval (reference :154-160) does not depend on the one-step-late decision.
        next_step = global_step.value() + 1
        inflight = []
        last_t = time.time()
        submitting = True

        def may_submit():
            # only state that is identical on every rank (agreed flags, step counters): the ranks must run the same sequence of
            # agreements; what a rank knows alone (its coordinator, its feeder) goes INTO the agreement below
            if not submitting or len(inflight) >= 2:
                return False
            return not (args.max_steps and (next_step > args.max_steps or steps_run + len(inflight) >= args.max_steps))

        while True:
            while may_submit():
                # replicas agree BEFORE a step is enqueued (it contains the gradient all-reduce: a rank that enqueued it while
                # another rank has no batch would wait for that rank forever)
                failure, have = 0, False
                if not coord.should_stop():
                    try:
                        have = model.next_batch_ready()
                    except Exception as e:      # feeder / staging error on this rank
                        log('Exiting due to exception: %s' % e)
                        traceback.print_exc()
                        coord.request_stop(e)
                        failure = 1
                failure, missing = agree(failure, 0 if have else 1)
                if failure or missing:
                    submitting = False
                    break
                t = model.submit_step(snapshot=(next_step % args.checkpoint_interval == 0))
                t.step_number = next_step
                next_step += 1
                inflight.append(t)
            if not inflight:
                break
            t = inflight.pop(0)
            err_flag = 0
            try:
                step, loss, _, loss_regularity = model.collect(t)
            except RuntimeError as e:           # hand-off timeout on this rank: the optimizer skipped the step on the device
                log('Exiting due to exception: %s' % e)
                coord.request_stop(e)
                err_flag, step, loss, loss_regularity = 1, t.step_number, float('nan'), 0.0
            now = time.time()
            time_window.append(now - last_t)
            last_t = now
            loss_window.append(loss)
            log('Step %-7d [%.03f avg_sec/step,  loss=%.05f,  avg_loss=%.05f,  lossw=%.05f]' % (
                step, time_window.average, loss, loss_window.average, loss_regularity))
            # if the gradient seems to explode, then restore to the previous step (reference :154-160).  Under data
            # parallelism the replicas must take the same branch: the decision is the MAX over ranks of [error, spike] (each
            # rank only sees the loss of its own shard), and a rank reads a checkpoint only after rank 0 has finished writing it.
            spike = 1 if (loss > 2 * loss_window.average or math.isnan(loss)) else 0
            err_flag, spike = agree(err_flag, spike)
            if err_flag:
                break
            if step % args.summary_interval == 0:
                # the scalars of reference add_stats (:23-36), from values the step already holds (TF histograms: out of scope)
                log('Summary at step %d: loss_mel=%.05f  loss_linear=%.05f  learning_rate=%.3e  max_gradient_norm=%.05f' % (
                    step, model.mel_loss, model.linear_loss, model.learning_rate, model.max_gradient_norm))
            if spike:
                log('recover to the previous checkpoint')
                restore_step = int((step - 10) / args.checkpoint_interval) * args.checkpoint_interval
                restore_path = '%s-%d' % (checkpoint_path, restore_step)
                # steps submitted behind the spike are discarded with it (the restore is ordered behind them on the stream);
                # a missing checkpoint raises (like saver.restore at reference :159) instead of training on with NaN weights
                inflight = []
                model.load_state_dict(torch.load(restore_path, weights_only=True))
                next_step = restore_step + 1
                saved = [p for p in saved if int(p.rsplit('-', 1)[1]) <= restore_step]
                steps_run += 1
                continue
            if step % args.checkpoint_interval == 0:
                if rank == 0:
                    path = '%s-%d' % (checkpoint_path, step)
                    log('Saving checkpoint to: %s' % path)
                    torch.save(t.state_dict(), path)        # the state cloned right behind step `step` (later steps are in flight)
                    saved.append(path)
                    while len(saved) > 5:                   # tf.train.Saver(max_to_keep=5, keep_checkpoint_every_n_hours=8)
                        old = saved.pop(0)
                        if os.path.getmtime(old) - kept_forever_at >= 8 * 3600:
                            kept_forever_at = os.path.getmtime(old)
                        elif os.path.exists(old):
                            os.remove(old)
                if world > 1:
                    dist.barrier(group=ctl)     # the file is complete before any rank may roll back to it
            t.snapshot = None
            steps_run += 1
    except Exception as e:
        log('Exiting due to exception: %s' % e)
        traceback.print_exc()
        coord.request_stop(e)
    finally:
        coord.request_stop()
        model.stop()
        if world > 1:
            try:                                  # leave the process group in order (its threads must not outlive the interpreter)
                torch.cuda.synchronize()
                dist.destroy_process_group()
            except Exception:                     # noqa: BLE001 -- a failed peer may already be gone
                pass


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
