"""Trainer / evaluator -- mirror of /root/reference/code/homography_CNN_synthetic.py.

Keeps the reference's CLI flags (:49-85), optimizer and learning-rate law (:161-183: Adam,
exponential_decay(lr, step, int(decay_steps), 0.96, staircase) with
decay_steps = ln(.96)*150000/ln(min_lr/lr)), the per-tower model construction (:229-257), gradient
averaging (:277-278 -> dist.GradAverager over RCCL) and the test-mode statistics (:391-580), as an
eager one-process-per-GPU program.  Inputs: with --data_path, the reference's on-disk layout through dataloader.Dataloader
(decode worker processes + uh_prepare_inputs: dataloader.py:76-235, joint augmentation in train mode, disjoint in test mode);
without it, synthetic.make_batch draws pairs with the same sampling law directly in HBM.

    python -m unsuperviseddeephomographyral2018_amd.homography_CNN_synthetic --mode train --batch_size 64 \
        --loss_type l1_loss --num_total_steps 200
    torchrun --nproc-per-node 8 -m unsuperviseddeephomographyral2018_amd.homography_CNN_synthetic --batch_size 512
"""
import argparse
import math
import os
import time

import numpy as np
import torch

from . import dataloader as uh_data
from . import dist as uh_dist
from . import synthetic
from .homography_model import HomographyModel, VGGRegressor, homography_model_params

# Size of synthetic image and the pertubation range (RH0)      (:14-17)
HEIGHT = 240
WIDTH = 320
RHO = 45
PATCH_SIZE = 128
AUGMENT_LIST = ['normalize']


def str2bool(s):
    return s.lower() == 'true'


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--mode', type=str, default='train', help='Train or test', choices=['train', 'test'])
    p.add_argument('--loss_type', type=str, default='l1_loss', help='Loss type',
                   choices=['h_loss', 'rec_loss', 'ssim_loss', 'l1_loss', 'l1_smooth_loss', 'ncc_loss'])
    p.add_argument('--use_batch_norm', type=str2bool, default='False', help='Use batch_norm?')
    p.add_argument('--leftright_consistent_weight', type=float, default=0)
    p.add_argument('--augment_list', nargs='+', default=AUGMENT_LIST, help='List of augmentations')
    p.add_argument('--do_augment', type=float, default=0.5)
    p.add_argument('--num_gpus', type=int, default=None,
                   help='Number of splits (towers) of the batch.  One process per GPU here: without a launcher, N > 1 re-executes '
                        'this program under torch.distributed.run with N ranks (RCCL when N GPUs are visible, gloo sharing the '
                        'visible ones otherwise); under a launcher it must equal WORLD_SIZE (error otherwise).  Not given = the '
                        'launcher decides (WORLD_SIZE, else 1) -- the reference\'s default of 2 would fail on a 1-GPU box')
    p.add_argument('--log_dir', type=str, default='../logs/')
    p.add_argument('--results_dir', type=str, default='../results/synthetic/report/')
    p.add_argument('--model_dir', type=str, default='../models/synthetic_models')
    p.add_argument('--model_name', type=str, default='model.ckpt')
    # dataset in the reference's on-disk layout (:62-70); empty data_path -> in-HBM synthetic pairs (synthetic.py)
    p.add_argument('--data_path', type=str, default='', help='The raw data path (I/, I_prime/ under it)')
    p.add_argument('--pts1_file', type=str, default='', help='4 corners on the first image - training dataset')
    p.add_argument('--test_pts1_file', type=str, default='')
    p.add_argument('--gt_file', type=str, default='', help='The training ground truth file')
    p.add_argument('--test_gt_file', type=str, default='')
    p.add_argument('--filenames_file', type=str, default='', help='File that contains all names of files, for training')
    p.add_argument('--test_filenames_file', type=str, default='')
    # accepted for command-line compatibility (:64-65, :73-74).  --I_dir / --I_prime_dir are parsed and never read by the
    # reference either (its Dataloader joins data_path + 'I/' / 'I_prime/', dataloader.py:143-144); --visual drives its interactive
    # matplotlib side (:117, :334, :362-387, :559-567), which is outside the hot path: main() says so once.  --save_visual is
    # honoured in test mode (one correspondence image per step into --results_dir, :539-552, drawn with PIL)
    p.add_argument('--I_dir', type=str, default='', help='The training image path (unused, as in the reference: data_path + I/)')
    p.add_argument('--I_prime_dir', type=str, default='', help='The training image path (unused: data_path + I_prime/)')
    p.add_argument('--visual', type=str2bool, default='false', help='Visualize obtained images to debug (accepted; no plotting here)')
    p.add_argument('--save_visual', type=str2bool, default='True', help='Save visual images for report: test mode writes one correspondence image per step into --results_dir')
    p.add_argument('--img_w', type=int, default=WIDTH)
    p.add_argument('--img_h', type=int, default=HEIGHT)
    p.add_argument('--patch_size', type=int, default=PATCH_SIZE)
    p.add_argument('--batch_size', type=int, default=128)
    p.add_argument('--max_epoches', type=int, default=150)
    p.add_argument('--lr', type=float, default=1e-4, help='Max learning rate')
    p.add_argument('--min_lr', type=float, default=.9e-4, help='Min learning rate')
    p.add_argument('--resume', type=str2bool, default='False')
    p.add_argument('--retrain', type=str2bool, default='False')
    # --- additions (not in the reference) ---
    p.add_argument('--rho', type=int, default=RHO)
    p.add_argument('--num_total_steps', type=int, default=150000, help='the reference hard-codes 150000 (:159)')
    p.add_argument('--num_test_data', type=int, default=1024, help='synthetic stand-in for the test file list')
    p.add_argument('--fused_patch', type=str2bool, default='False',
                   help='l1_loss only: fused patch kernel instead of the full-frame warp')
    p.add_argument('--solve_f64', type=str2bool, default='False', help='carry the 8x8 DLT solve in f64')
    p.add_argument('--zero_nonfinite_grad', type=str2bool, default='True',
                   help='train mode: a pair whose d loss / d pred_h4p is NaN / Inf (degenerate predicted corners) contributes '
                        'no gradient instead of poisoning every variable (UH_DLT_ZERO_NONFINITE_GRAD).  False = the '
                        'reference\'s behaviour: tf.matrix_solve + autodiff pass the NaN on.  The log line counts the pairs.')
    p.add_argument('--tf_adam_epsilon', type=str2bool, default='True',
                   help='Adam update as tf.train.AdamOptimizer applies it (epsilon = the paper\'s "epsilon hat": homography_CNN_synthetic.tf_adam_eps); '
                        'False = torch.optim.Adam\'s constant eps (rounds 1-5).  --step_graph always uses the latter')
    p.add_argument('--tunable_gemm', type=str2bool, default='True',
                   help='PyTorch TunableOp for the fully connected GEMMs (dist.tune_gemms): the first call of each GEMM shape benchmarks '
                        'the rocBLAS / hipBLASLt candidates, as cudnn.benchmark does for the convs (~3 s once per host; fc1 forward '
                        '142 -> 55 us, the step -1.5 %)')
    p.add_argument('--step_graph', type=str2bool, default='False',
                   help='capture the whole training step (convs, hot path, Adam) into one hipGraph and replay it')
    p.add_argument('--graph_tail', type=str2bool, default='False',
                   help='l1_loss only: DLT -> warp -> loss and their backward as one library call / one hipGraph launch')
    p.add_argument('--fresh_data_every', type=int, default=1, help='draw a new synthetic batch every N steps')
    p.add_argument('--texture', type=str, default='smooth', choices=['smooth', 'multiscale', 'white'],
                   help='in-HBM synthetic image texture (synthetic.py); ignored with --data_path')
    p.add_argument('--data_pool', type=int, default=0,
                   help='> 0: pre-generate this many synthetic batches in HBM (118 MB each at 64 x 240x320: 288 GB holds '
                        'thousands) and cycle through them in a fresh random order per pass, instead of generating per step')
    p.add_argument('--decode_workers', type=int, default=-1,
                   help='--data_path: image-decode worker PROCESSES (dataloader.Dataloader(num_workers=...)); 0 = the 20-thread '
                        'pool (interpreter-lock bound at ~2 000 pairs/s); -1 = min(16, host cores / (4 x ranks)) per rank')
    p.add_argument('--exchange_report', type=str2bool, default='True',
                   help='world > 1: after training, print what the gradient exchange cost (dist.exchange_report; ~25 extra steps, '
                        'the trained state is restored afterwards)')
    p.add_argument('--test_shuffle', type=str2bool, default='False',
                   help='test mode with --data_path: True = shuffle the test list as the reference does (Dataloader(..., shuffle=True), '
                        ':404); default: walk it in order -- every pair exactly 3 times, reproducible statistics')
    p.add_argument('--seed', type=int, default=0)
    p.add_argument('--log_every', type=int, default=100)
    p.add_argument('--save_every', type=int, default=1000)
    return p


def decay_steps_for(lr, min_lr, num_total_steps=150000, decay_rate=0.96):
    """(:166) -- note the reference always uses 150000 here, whatever the real step count."""
    return (math.log(decay_rate) * num_total_steps) / math.log(min_lr * 1.0 / lr)


def staircase_lr(lr, step, decay_steps, decay_rate=0.96):
    """tf.train.exponential_decay(lr, step, int(decay_steps), decay_rate, staircase=True)  (:169)"""
    return lr * decay_rate ** (step // int(decay_steps))


def tf_adam_eps(t, eps_hat=1e-8, beta2=0.999):
    """The `eps` that makes torch.optim.Adam's t-th update tf.train.AdamOptimizer's (the reference's optimizer, :183).  TF1 applies
    lr_t * m / (sqrt(v) + eps_hat) with lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) -- its epsilon is the paper's "epsilon hat";
    torch applies lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps).  The two coincide iff
    eps = eps_hat / sqrt(1 - beta2^t): 31.6 eps_hat at t = 1, -> eps_hat as t grows.  (With a constant eps = 1e-8 torch's effective
    epsilon-hat is up to 31.6 x smaller; it only matters for variables whose gradients are ~1e-8, but it is the reference's rule.)"""
    return eps_hat / math.sqrt(1.0 - beta2 ** int(t))


class TrainStep(object):
    """One tower: shared variables + Adam + (for world > 1) overlapped RCCL gradient averaging."""

    def __init__(self, args, device, world=1, net=None):
        self.args = args
        self.device = device
        self.world = world
        self.net = net if net is not None else VGGRegressor(args.patch_size, args.use_batch_norm)
        self.net = self.net.to(device).to(memory_format=torch.channels_last)
        self.rank = torch.distributed.get_rank() if (world > 1 and torch.distributed.is_initialized()) else 0
        if world > 1:                                       # identical initial variables on every rank
            for t in list(self.net.parameters()) + list(self.net.buffers()):
                torch.distributed.broadcast(t.data, src=0)
            # ... and independent dropout masks per tower, as the reference's one-dropout-op-per-tower graph draws them
            # (homography_model.py:120-121,128; SURVEY 8e "Dropout RNG per rank").  World size 1 keeps the caller's seed.
            self.model_rng_seed = uh_dist.seed_tower_rng(getattr(args, 'seed', 0), self.rank)
        self.decay_steps = decay_steps_for(args.lr, args.min_lr)
        self.global_step = 0
        self.adam_t = 0                                    # Adam updates applied so far (TF's beta-power accumulators; survives --retrain)
        # world == 1: no exchange step -> no flat buckets: autograd hands each parameter its gradient directly (no
        # 137 MB zero fill + read-modify-write accumulation per step)
        self.averager = uh_dist.GradAverager(self.net, world) if world > 1 else None
        kw = {}
        self.step_graph = bool(getattr(args, 'step_graph', False)) and device.type == 'cuda' and world == 1
        if getattr(args, 'step_graph', False) and world > 1:
            import warnings
            warnings.warn('--step_graph is ignored with world size %d: the RCCL all-reduces are issued from autograd hooks '
                          'and are not captured; running the eager step' % world)
        # bench / test hook: per-pair corner offsets ADDED to the regressor's output (None = off).  bench.py sets
        # gt + N(0, 2 px) so that theta in the timed steps follows SURVEY section 8(d)'s mid-training law instead of the
        # near-identity a 25-step-old regressor predicts; the step itself (conv fwd/bwd, hot path, Adam) is unchanged.
        self.h4p_offset = None
        lr0 = args.lr
        if device.type == 'cuda':
            kw['fused'] = True
        if self.step_graph:
            # whole-step hipGraph: Adam must not read host scalars at replay time -> capturable state, lr as a tensor
            kw['capturable'] = True
            lr0 = torch.tensor(float(args.lr), device=device)
        self.opt = torch.optim.Adam([p for p in self.net.parameters() if p.requires_grad], lr=lr0,
                                    betas=(0.9, 0.999), eps=1e-8, **kw)
        self._graph = None
        # the reference's Adam is TF1's ("epsilon hat" form): torch's eps is set per step so that the update is the same (tf_adam_eps).
        # Not in the whole-step hipGraph, where Adam's eps is a captured constant: that mode keeps torch's rule.
        self.tf_adam_epsilon = bool(getattr(args, 'tf_adam_epsilon', True)) and not self.step_graph
        self.model_params = homography_model_params(
            mode=args.mode, batch_size=int(args.batch_size / world), patch_size=args.patch_size,
            img_h=args.img_h, img_w=args.img_w, loss_type=args.loss_type, use_batch_norm=args.use_batch_norm,
            augment_list=args.augment_list, leftright_consistent_weight=args.leftright_consistent_weight)

    def learning_rate(self):
        return staircase_lr(self.args.lr, self.global_step, self.decay_steps)

    # ---- whole-step hipGraph (torch.cuda.CUDAGraph on ROCm) -----------------------------------------------
    # One training step is ~150 launches (convs, epilogues, the hot-path kernels, fused Adam).  The eager stream is already
    # GPU-bound -- 5 492 of 5 512 us busy, 0.4 % in launch gaps (profiles/r04_step_breakdown.txt) -- so the graph buys no
    # throughput at batch 64; it exists for small batches / slow hosts, where the ~150 launches per step outrun the GPU.
    # Shapes are static, so the step is captured ONCE into a hipGraph over static input buffers and replayed: one graph
    # launch per step.  The library's kernels are captured like any other stream work (they only enqueue on the stream).
    def _set_lr(self):
        lr = self.learning_rate()
        for g in self.opt.param_groups:
            if torch.is_tensor(g['lr']):
                g['lr'].fill_(lr)
            else:
                g['lr'] = lr

    def _capture(self, batch):
        import copy
        self._static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        # The eager warm-up below (MIOpen find, allocator, Adam state creation) must not count as training: snapshot the
        # variables, the optimizer state and the RNG streams (dropout) and put them back, so that step k of a
        # --step_graph run sees exactly what step k of an eager run sees -- also when the state was just restored from a
        # checkpoint (--resume): the Adam moments and step counts are copied back, not zeroed.
        net_sd = copy.deepcopy(self.net.state_dict())
        opt_before = {p: {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                      for p, st in self.opt.state.items()}
        rng_cpu, rng_dev = torch.get_rng_state(), torch.cuda.get_rng_state(self.device)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(3):
                self._forward_backward_update(self._static)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        self.net.load_state_dict(net_sd)
        with torch.no_grad():        # Adam: keep the (capturable) state TENSORS the graph will update, restore their contents
            for p, st in self.opt.state.items():
                old = opt_before.get(p)
                for k, v in st.items():
                    if torch.is_tensor(v):
                        if old is not None and torch.is_tensor(old.get(k)):
                            v.copy_(old[k])
                        else:
                            v.zero_()                    # the state did not exist before the warm-up: a fresh optimizer
        torch.set_rng_state(rng_cpu); torch.cuda.set_rng_state(rng_dev, self.device)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._graph_model = self._forward_backward_update(self._static)

    def _zero_nonfinite(self):
        return bool(getattr(self.args, 'zero_nonfinite_grad', True)) and self.args.mode == 'train'

    def _zero_or_reset(self):
        if self.averager is not None:
            self.averager.reset()
        else:
            self.opt.zero_grad(set_to_none=True)

    def prime_conv_finds(self, batch):
        """World > 1, before the first step: rank 0 ALONE runs one forward + backward (no collective, no optimizer step)
        so that MIOpen's find mode (torch.backends.cudnn.benchmark: ~24 conv problems, tens of seconds of kernel builds)
        happens ONCE and lands in the user find-db / kernel cache; the other ranks run the same pass after a barrier and hit
        what rank 0 stored.  N concurrent finds all build the same kernels and contend for one sqlite kernel cache and one
        find-db file.  UH_FIND_STAGGER=0 switches the staggering off (every rank primes at once).  Gradients are discarded;
        every rank draws one step's worth of dropout masks."""
        if self.world == 1:
            return 0.0
        stagger = os.environ.get('UH_FIND_STAGGER', '1') != '0'
        t0 = time.perf_counter()

        def one_pass():
            self.averager.enabled = False
            try:
                self.averager.reset()
                model = HomographyModel(self.model_params, *synthetic.model_args(batch), reuse_variables=True,
                                        net=self.net, fused_patch=self.args.fused_patch, solve_f64=self.args.solve_f64,
                                        h4p_offset=self.h4p_offset, zero_nonfinite_grad=self._zero_nonfinite())
                model.loss.backward()
                if self.device.type == 'cuda':
                    torch.cuda.synchronize(self.device)
            finally:
                self.averager.enabled = True
                self.averager.reset()
        if stagger:
            if self.rank == 0:
                one_pass()
            torch.distributed.barrier()
            if self.rank != 0:
                one_pass()
        else:
            one_pass()
        torch.distributed.barrier()
        return time.perf_counter() - t0

    def _forward_backward_update(self, batch):
        self._zero_or_reset()
        model = HomographyModel(self.model_params, *synthetic.model_args(batch), reuse_variables=True,
                                net=self.net, fused_patch=self.args.fused_patch, solve_f64=self.args.solve_f64,
                                h4p_offset=self.h4p_offset, zero_nonfinite_grad=self._zero_nonfinite())
        model.loss.backward()
        if self.averager is not None:
            self.averager.finish()
        self.opt.step()
        return model

    def __call__(self, batch):
        """sess.run([apply_grad_opt, ...]) of the hot loop (:333-353): forward, backward, average, Adam."""
        if self.step_graph:
            self._set_lr()
            if self._graph is None:
                self._capture(batch)                     # its eager warm-up steps are rolled back
            for k, v in batch.items():                   # new data lands in the static buffers the graph reads
                if torch.is_tensor(v) and v.data_ptr() != self._static[k].data_ptr():
                    self._static[k].copy_(v)
            self._graph.replay()
            self.global_step += 1
            self.adam_t += 1
            return self._graph_model                     # its tensors are the graph's static outputs
        if getattr(self.args, 'graph_tail', False) and self.device.type == 'cuda':
            # stream capture needs a real (non-NULL) stream: run the whole step on a private one
            if not hasattr(self, '_stream'):
                self._stream = torch.cuda.Stream(device=self.device)
            self._stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self._stream):
                model = self._step(batch)
            torch.cuda.current_stream(self.device).wait_stream(self._stream)
            return model
        return self._step(batch)

    def _step(self, batch):
        lr = self.learning_rate()
        for g in self.opt.param_groups:
            g['lr'] = lr
            if self.tf_adam_epsilon:
                g['eps'] = tf_adam_eps(self.adam_t + 1, 1e-8, g['betas'][1])
        self._zero_or_reset()
        model = HomographyModel(self.model_params, *synthetic.model_args(batch), reuse_variables=True,
                                net=self.net, fused_patch=self.args.fused_patch, solve_f64=self.args.solve_f64,
                                graph_tail=getattr(self.args, 'graph_tail', False), h4p_offset=self.h4p_offset,
                                zero_nonfinite_grad=self._zero_nonfinite())
        model.loss.backward()
        if self.averager is not None:
            self.averager.finish()
        self.opt.step()
        self.global_step += 1
        self.adam_t += 1
        return model

    def state_dict(self):
        return {'net': self.net.state_dict(), 'opt': self.opt.state_dict(), 'global_step': self.global_step, 'adam_t': self.adam_t}

    def load_state_dict(self, sd, retrain=False):
        self.net.load_state_dict(sd['net'])
        self.opt.load_state_dict(sd['opt'])
        # Optimizer.load_state_dict also restores the SAVED run's param-group options; whether Adam is capturable and holds
        # its lr in a device tensor is a property of THIS run (--step_graph), not of the checkpoint
        for g in self.opt.param_groups:
            g['capturable'] = self.step_graph
            lr = float(g['lr'])
            g['lr'] = torch.tensor(lr, device=self.device) if self.step_graph else lr
        for st in self.opt.state.values():                  # fused / capturable Adam keeps `step` as a device f32 tensor
            if torch.is_tensor(st.get('step')) and self.device.type == 'cuda':
                st['step'] = st['step'].to(device=self.device, dtype=torch.float32)
        self.global_step = 0 if retrain else sd['global_step']        # (:314-317)
        # TF restores Adam's beta-power accumulators with the other variables: --retrain resets the step counter, not them
        if 'adam_t' in sd:
            self.adam_t = int(sd['adam_t'])
        else:                                                          # a checkpoint of rounds 1-5: read it from the optimizer state
            steps = [st['step'] for st in self.opt.state.values() if 'step' in st]
            self.adam_t = int(float(steps[0])) if steps else 0
        # a step graph captured BEFORE this call replays against the old lr / state tensors: capture again on the next step
        self._graph = None
        self._static = None


def _ckpt_path(args):
    prefix = args.loss_type + ''.join('_' + a for a in args.augment_list)
    d = os.path.join(args.model_dir, prefix)
    return d, os.path.join(d, args.model_name + '.pt')


def train(args):
    rank, world, local = uh_dist.init_from_env()
    device = torch.device('cuda', local) if torch.cuda.is_available() else None
    if device is None:
        raise RuntimeError('training needs an MI355X: the hot path has no CPU fallback')
    torch.cuda.set_device(device)
    torch.backends.cudnn.benchmark = True                   # MIOpen find: shapes are static
    uh_dist.skip_naive_conv_in_find()
    if getattr(args, 'tunable_gemm', True):
        uh_dist.tune_gemms()                                # ... and the same kind of search for the fully connected GEMMs
    torch.manual_seed(args.seed)
    step_fn = TrainStep(args, device, world)
    ckpt_dir, ckpt = _ckpt_path(args)
    if args.resume and os.path.exists(ckpt):
        step_fn.load_state_dict(torch.load(ckpt, map_location=device), retrain=args.retrain)
    start_step = step_fn.global_step
    if rank == 0:
        print('===> Decay steps:', step_fn.decay_steps)
        print('===> Start step:', start_step, ' world size:', world)
    B_local = args.batch_size // world
    names = ['h_loss', 'rec_loss', 'ssim_loss', 'l1_loss', 'l1_smooth_loss', 'ncc_loss']
    batch = None
    disk = None
    if args.data_path:
        # the reference's Dataloader (:186) on its on-disk format; every rank reads its own shuffled stream
        prm = uh_data.dataloader_params(data_path=args.data_path, filenames_file=args.filenames_file,
                                        pts1_file=args.pts1_file, gt_file=args.gt_file, mode='train', batch_size=B_local,
                                        img_h=args.img_h, img_w=args.img_w, patch_size=args.patch_size,
                                        augment_list=args.augment_list, do_augment=args.do_augment)
        # per rank; measured on a 256-thread host: 16 workers feed 11.7 k pairs/s, 48 only 9.6 k (they crowd the trainer's thread)
        nw = args.decode_workers if args.decode_workers >= 0 else max(2, min(16, (os.cpu_count() or 8) // (4 * world)))
        loader = uh_data.Dataloader(prm, shuffle=True, device=device, seed=args.seed * 1000 + rank, num_workers=nw)

        disk = loader.stream(prefetch=4)       # endless, never drops a pair, raises on an empty list; decode overlaps the step
    pool, order = [], []
    if disk is None and args.data_pool > 0:
        pool = [synthetic.make_batch(B_local, args.img_h, args.img_w, args.patch_size, args.rho,
                                     seed=args.seed + i * world + rank, device=device, kind=args.texture)
                for i in range(args.data_pool)]
        pool_gen = torch.Generator().manual_seed(args.seed * 7919 + rank)
    if world > 1:
        # MIOpen find mode once, by rank 0, before anyone steps (TrainStep.prime_conv_finds); any batch of the step's shape does
        t_find = step_fn.prime_conv_finds(synthetic.make_batch(B_local, args.img_h, args.img_w, args.patch_size, args.rho,
                                                               seed=args.seed + rank, device=device, kind=args.texture))
        if rank == 0:
            print('===> conv find pass (rank 0 first, then the others): %.1f s' % t_find, flush=True)
    monitors = uh_dist.TowerMonitors(len(names), world, device)
    from . import _lib
    _lib.dlt_zeroed_pairs(reset=True)
    zeroed_pairs = 0
    t0 = time.time()
    try:
        for step in range(start_step, start_step + args.num_total_steps):
            if disk is not None:
                batch = next(disk)
            elif pool:
                if not order:
                    order = torch.randperm(len(pool), generator=pool_gen).tolist()
                batch = pool[order.pop()]
            elif batch is None or (step - start_step) % max(args.fresh_data_every, 1) == 0:
                # every rank draws its own shard (seeded by step and rank) == tf.split of a global batch
                batch = synthetic.make_batch(B_local, args.img_h, args.img_w, args.patch_size, args.rho,
                                             seed=args.seed + step * world + rank, device=device, kind=args.texture)
            model = step_fn(batch)
            # Monitors (:333-352): per-rank running sums on the device, ONE collective per log line (dist.TowerMonitors)
            monitors.add([getattr(model, n) for n in names])
            if step % args.log_every == 0:
                z = _lib.dlt_zeroed_pairs(reset=True)            # (synchronous read of a device counter: log time only)
                mon = monitors.reduce(extra_count=z)
                zeroed_pairs += mon['extra']
                if rank == 0:
                    n = step - start_step + 1
                    avg, win = mon['avg'], mon['win']            # running means since the start, as the reference prints (:345-352),
                    dt = time.time() - t0                        # and the mean over the steps since the previous log line
                    print('Train: step %d  ' % step + ', '.join('%s %.6f' % (kk, v) for kk, v in zip(names, avg))
                          + ', lr %.6f, %.1f pairs/s' % (step_fn.learning_rate(), n * args.batch_size / max(dt, 1e-9))
                          + '  | last %d steps: h_loss %.4f %s %.6f' % (mon['win_n'], win[0], args.loss_type, win[names.index(args.loss_type)])
                          + ('  | %d tower-steps with a non-finite loss value so far' % mon['nonfinite'] if mon['nonfinite'] else '')
                          + ('  | %d pairs had a non-finite gradient zeroed so far (--zero_nonfinite_grad; the reference would have '
                             'propagated NaN)' % zeroed_pairs if zeroed_pairs else ''),
                          flush=True)
            if rank == 0 and step and step % args.save_every == 0:
                os.makedirs(ckpt_dir, exist_ok=True)
                torch.save(step_fn.state_dict(), ckpt)
    finally:
        if disk is not None:
            disk.close()             # ends the producer thread, the decode workers and the page-locked frame ring -- also on
                                     # an exception or Ctrl-C (the ring used to outlive the process)
    if rank == 0:
        os.makedirs(ckpt_dir, exist_ok=True)
        torch.save(step_fn.state_dict(), ckpt)          # Save the final model (:389)
    if world > 1 and batch is not None and getattr(args, 'exchange_report', True):
        # self-diagnosing multi-GPU run (dist.exchange_report): what the gradient exchange cost, from extra steps AFTER training
        # on the last batch.  Those steps are real optimizer steps: the trained state is snapshotted and put back on every rank.
        import copy
        import json
        snap = copy.deepcopy(step_fn.state_dict())

        def run_steps(n):
            for _ in range(n):
                step_fn(batch)
        rep = uh_dist.exchange_report(step_fn.averager, run_steps, ms_per_step=None)
        step_fn.load_state_dict(snap)
        if rank == 0:
            print('===> exchange (world %d, %s): %s' % (world, torch.distributed.get_backend(), json.dumps(rep)), flush=True)
    return step_fn


def denorm_img(img):
    """utils.denorm_img (utils/utils.py:406-413): x * std + mean with the dataloader's constants (per channel, or their means for
    a gray image)."""
    mean, std = np.array(synthetic.MEAN_I), np.array(synthetic.STD_I)
    if img.ndim == 2:
        mean, std = mean.mean(), std.mean()
    return img * std + mean


def save_correspondences_img(img1, img2, corr1, corr2, pred_corr2, results_dir, img_name):
    """utils.save_correspondences_img + draw_matches for four points (utils/utils.py:209-224,230-308) with PIL instead of cv2: the two
    uint8 frames side by side, the predicted quadrilateral on the second frame, both ground-truth quadrilaterals, and the four
    correspondences joined by coloured lines with a circle of radius 7 at each end.  -> the path written."""
    from PIL import Image, ImageDraw
    img1, img2 = np.asarray(img1, np.uint8), np.asarray(img2, np.uint8)
    h = max(img1.shape[0], img2.shape[0])
    canvas = np.zeros((h, img1.shape[1] + img2.shape[1], 3), np.uint8)
    canvas[:img1.shape[0], :img1.shape[1]] = img1
    canvas[:img2.shape[0], img1.shape[1]:] = img2
    im = Image.fromarray(canvas)
    d = ImageDraw.Draw(im)
    off = np.array([img1.shape[1], 0])
    quad = lambda pts: [tuple(int(v) for v in p) for p in np.asarray(pts).reshape(4, 2)] + [tuple(int(v) for v in np.asarray(pts).reshape(4, 2)[0])]
    d.line(quad(np.asarray(pred_corr2).reshape(4, 2) + off), fill=(5, 225, 225), width=3)                # the prediction (:214)
    d.line(quad(np.asarray(corr2).reshape(4, 2) + off), fill=(2, 10, 240), width=3)                     # ground truth, second frame (:275)
    d.line(quad(corr1), fill=(2, 10, 240), width=3)                                                     # ... first frame (:276)
    colors = [(255, 102, 255), (51, 153, 255), (102, 255, 255), (255, 255, 0)]                          # line_color_set (:217)
    for k, (a, b) in enumerate(zip(np.asarray(corr1).reshape(4, 2), np.asarray(corr2).reshape(4, 2))):
        e1 = tuple(int(v) for v in np.round(a)); e2 = tuple(int(v) for v in np.round(b) + off)
        d.line([e1, e2], fill=colors[k], width=1)
        for e in (e1, e2):
            d.ellipse([e[0] - 7, e[1] - 7, e[0] + 7, e[1] + 7], outline=colors[k], width=1)
    os.makedirs(results_dir, exist_ok=True)
    path = os.path.join(results_dir, img_name)
    im.save(path, quality=95)
    return path


class TestHomography(object):
    """Test-mode loop (:391-580): mean corner error (bounded RMSE), failure rate and percentiles."""

    def __init__(self, args, step_fn=None):
        self.args = args
        device = torch.device('cuda', 0)
        args_t = argparse.Namespace(**vars(args)); args_t.mode = 'test'
        self.step_fn = step_fn
        net = step_fn.net if step_fn is not None else VGGRegressor(args.patch_size, args.use_batch_norm).to(device)
        if step_fn is None:
            _, ckpt = _ckpt_path(args)
            if not os.path.exists(ckpt):
                # the reference fails on restore (saver.restore, :427-431); statistics of a random net are not results
                raise FileNotFoundError('TestHomography: no checkpoint at %s (train first, or pass a TrainStep)' % ckpt)
            net.load_state_dict(torch.load(ckpt, map_location=device)['net'])
            net = net.to(memory_format=torch.channels_last)
        self.net = net
        self.device = device
        self.params = homography_model_params(
            mode='test', batch_size=args.batch_size, patch_size=args.patch_size, img_h=args.img_h,
            img_w=args.img_w, loss_type=args.loss_type, use_batch_norm=args.use_batch_norm,
            augment_list=args.augment_list, leftright_consistent_weight=args.leftright_consistent_weight)

    def run(self, save_visual=False):
        """save_visual: write one correspondence image per test step into args.results_dir, as the reference's test loop does with
        --save_visual True (:539-552).  The command line passes the flag; library callers (bench.py) leave it off."""
        a = self.args
        disk = None
        num_test_data = a.num_test_data
        if getattr(a, 'data_path', '') and getattr(a, 'test_filenames_file', ''):
            # the reference's test Dataloader (:138-148, :404): disjoint augmentation.  The reference passes shuffle=True there (its
            # "# No shuffle" comment is wrong), so its 3 epochs draw from a shuffling queue; the statistics are the same in
            # expectation.  Default here: the file list IN ORDER (every pair exactly three times, reproducible figures);
            # --test_shuffle True = the reference's shuffled stream
            prm = uh_data.dataloader_params(data_path=a.data_path, filenames_file=a.test_filenames_file,
                                            pts1_file=a.test_pts1_file, gt_file=a.test_gt_file, mode='test',
                                            batch_size=a.batch_size, img_h=a.img_h, img_w=a.img_w,
                                            patch_size=a.patch_size, augment_list=a.augment_list, do_augment=a.do_augment)
            loader = uh_data.Dataloader(prm, shuffle=bool(getattr(a, 'test_shuffle', False)), device=self.device, seed=a.seed)
            num_test_data = len(loader.names)
            if 0 < num_test_data < a.batch_size:                          # test_batch_size = min(n, batch_size) (:136)
                loader.params = prm._replace(batch_size=num_test_data)
                self.params = self.params._replace(batch_size=num_test_data)
                a = argparse.Namespace(**dict(vars(a), batch_size=num_test_data))

            disk = loader.stream()             # unshuffled: consecutive batches walk the list in order, wrapping
        steps_per_epoch = int(np.ceil(num_test_data / a.batch_size))
        num_steps = 3 * steps_per_epoch                                   # (:400-401)
        per_pair, per_step, total_fail, total_bounded = [], [], 0.0, 0.0
        was_training = self.net.training
        with torch.no_grad():
            for step in range(num_steps):
                batch = next(disk) if disk is not None else synthetic.make_batch(
                    a.batch_size, a.img_h, a.img_w, a.patch_size, a.rho, seed=10_000_000 + a.seed + step,
                    device=self.device, kind=getattr(a, 'texture', 'smooth'))
                m = HomographyModel(self.params, *synthetic.model_args(batch), reuse_variables=True, net=self.net,
                                    solve_f64=a.solve_f64)
                per_step.append(np.float32(float(m.bounded_h_loss)))       # h_losses_array.append(h_loss_value) (:537): float32, as fetched
                total_bounded += float(per_step[-1])
                total_fail += float(m.num_fail)
                per_pair.append(torch.sqrt(torch.mean((m.pred_h4p - m.gt) ** 2, dim=1)).cpu())
                if save_visual:                                            # the first pair of the batch (:539-552)
                    I_s = denorm_img(m.I[0].cpu().numpy()).clip(0, 255).astype(np.uint8)
                    Ip_s = denorm_img(m.I_prime[0].cpu().numpy()).clip(0, 255).astype(np.uint8)
                    p1 = m.pts_1[0].cpu().numpy().reshape(4, 2)
                    name = '%d_%s_loss_%s.jpg' % (step * a.batch_size, a.loss_type, per_step[-1])
                    save_correspondences_img(Ip_s, I_s, p1, p1 + m.gt[0].cpu().numpy().reshape(4, 2),
                                             p1 + m.pred_h4p[0].cpu().numpy().reshape(4, 2), a.results_dir, name)
        self.net.train(was_training)
        if disk is not None:
            disk.close()
        per_pair = torch.cat(per_pair).numpy()
        res = {
            'mean_corner_error': total_bounded / num_steps,
            'fail_percent': 100.0 * total_fail / (num_steps * a.batch_size),
            # the REFERENCE's printout (:577-579 -> utils.find_percentile, utils.py:655-672): [mean, std] of the sorted per-STEP
            # bounded h_loss values (one per test batch) in the intervals 0-30 %, 30-60 %, 60-100 %
            'reference_percentile_intervals': find_percentile(per_step).tolist(),
            # ... and two statistics the reference does not print (per PAIR, unbounded corner RMSE): point percentiles, and the
            # same three intervals -- the form the paper's bar charts use
            'percentiles': {q: float(np.percentile(per_pair, q)) for q in (20, 30, 50, 60, 80, 100)},
            'per_pair_intervals': find_percentile(per_pair).tolist(),
            'num_pairs': int(per_pair.size),
        }
        print('====> Result for RHO:', a.rho, ' loss ', a.loss_type, ' noise ', getattr(a, 'do_augment', None) if disk is not None else 0.0)
        print('|Average error: %.4f |Fail percent: %.3f' % (res['mean_corner_error'], res['fail_percent']))
        print('===> Percentile Values (the reference\'s find_percentile: [mean, std] of the per-batch h_loss in the sorted intervals '
              '0-30 %%, 30-60 %%, 60-100 %%): %s' % np.round(np.array(res['reference_percentile_intervals']), 3).tolist())
        print('===> per-pair corner RMSE (px; not a reference printout): point percentiles ' + ', '.join(
            '%d%%: %.3f' % (q, v) for q, v in sorted(res['percentiles'].items())) + '; intervals 0-30/30-60/60-100 %% [mean, std]: %s'
            % np.round(np.array(res['per_pair_intervals']), 3).tolist() + '  over %d pairs' % res['num_pairs'])
        return res


def find_percentile(x, tops_list=(0.3, 0.6, 1)):
    """utils.find_percentile (utils/utils.py:655-672): sort x; [mean, std] of the intervals [0, 30 %), [30 %, 60 %), [60 %, 100 %),
    in x's OWN dtype (the reference hands it the float32 values sess.run returned, so np.mean / np.std accumulate in float32);
    an empty interval gives [nan, nan] as np.mean / np.std of an empty slice do.  Pinned to the reference's function by
    tests/golden/ref_find_percentile.npz (tests/test_cli.py)."""
    import warnings
    xs = np.sort(x)
    out, start = [], 0
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)              # "Mean of empty slice": the reference prints it and goes on
        for t in tops_list:
            stop = int(t * len(xs))
            iv = xs[start:stop]
            out.append([np.mean(iv), np.std(iv)])
            start = stop
    return np.array(out)


MODULE = 'unsuperviseddeephomographyral2018_amd.homography_CNN_synthetic'


def unsupported_flag_notes(args, argv):
    """One line per reference flag that is accepted but has no effect here, only for flags the command line really set."""
    given = {a.split('=')[0] for a in argv if a.startswith('--')}
    notes = []
    if args.visual:
        notes.append('--visual True: the reference plots with matplotlib every step (:362-387, :559-567); there is no plotting side '
                     'here -- training / testing runs without it')
    for f in ('--I_dir', '--I_prime_dir'):
        if f in given:
            notes.append('%s: parsed and never read, exactly as in the reference (frames are data_path + I/ and I_prime/)' % f)
    return notes


def main(argv=None):
    import sys
    argv = list(sys.argv[1:] if argv is None else argv)
    args = build_parser().parse_args(argv)
    for note in unsupported_flag_notes(args, argv):
        if int(os.environ.get('RANK', '0')) == 0:
            print('note: ' + note, file=sys.stderr, flush=True)
    try:
        what, world = uh_dist.resolve_num_gpus(args.num_gpus, args.batch_size if args.mode == 'train' else None)
    except ValueError as e:
        raise SystemExit('homography_CNN_synthetic: %s' % e)
    if args.mode == 'train':
        if what == 'launch':
            if torch.cuda.is_available() and torch.cuda.device_count() < world:
                print('note: --num_gpus %d with %d visible GPU(s): the ranks share the visible device(s) over gloo (a functional '
                      'run, not a scaling run)' % (world, torch.cuda.device_count()), file=sys.stderr, flush=True)
            uh_dist.self_launch(world, MODULE, argv, module=True)          # does not return
        train(args)
    else:
        if world > 1 and int(os.environ.get('RANK', '0')) == 0:
            # the reference builds num_gpus towers in test mode too (:417-446) and averages their statistics: the figures do
            # not depend on the split, so the test loop runs on one device
            print('note: test mode evaluates on one GPU; --num_gpus / WORLD_SIZE %d does not change the statistics' % world,
                  file=sys.stderr, flush=True)
        if int(os.environ.get('RANK', '0')) == 0:
            TestHomography(args).run(save_visual=args.save_visual)


if __name__ == '__main__':
    main()
