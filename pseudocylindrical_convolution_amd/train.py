"""End-to-end training of the codec, one process per GPU over RCCL
(reference: test/trainDDP_Full.py; test/trainDDP_Base.py is the same loop with --base).

    python -m pseudocylindrical_convolution_amd.train --gpus 8 --data-dir ... --train-list ... --test-list ...
    python -m torch.distributed.run --nproc-per-node 8 -m pseudocylindrical_convolution_amd.train ...

The loop, the loss (gamma*viewport MSE + beta*(1 - viewport SSIM) + alpha*rate), the alternating
optimisers (entropy model / transforms + quantiser levels, the histogram "gradient" of `quant.count`
applied by its own SGD), gradient accumulation over `acc_batch` steps with clipping, and the
checkpoint naming follow the reference.  What differs: ranks come from the launcher's environment
(RANK / LOCAL_RANK / WORLD_SIZE) instead of mp.spawn, every path is an argument, and
`--synthetic N` / `--procedural N` train on generated images where no dataset exists, and
`--time-budget S` ends the run after S seconds of wall time (the epoch in flight stops at its next
step, is tested and saved as usual)."""
from __future__ import print_function

import argparse
import os
import sys
import time
from itertools import chain

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP

from . import model_zoo_v2
from .PCONV_operator import Logger, ModuleSaver, MultiProject, SSIM
from .RDMetric import mse_tb
from .SphereDataset import (ProceduralSphereDataSet, SphereDataSet, SyntheticSphereDataSet,
                            load_train_test_distribute)
from .model_zoo_v2 import AccGrad


def get_params(model, ent):
    m = model.module
    if ent:
        return m.ent.parameters()
    return chain(m.encoder.parameters(), m.decoder.parameters(), [m.quant.weight])


def forward_losses(args, model, data, pr1, pr2, sim_func):
    """(mse, ssim, rate) of one batch; rate is None for the base model"""
    out = model(data)
    y, ent_vec, mask = out if isinstance(out, tuple) else (out, None, None)
    py, px = pr1(y), pr2(data)
    mse = torch.mean((px - py) * (px - py))
    ssim = sim_func(px, py)
    rate = None if ent_vec is None else torch.sum(ent_vec) / torch.sum(mask).item()
    return mse, ssim, rate


def train(args, model, device, train_loader, optimizer, optimizer_quant, epoch, log, pr1, pr2, ent=True):
    """one epoch (reference: trainDDP_Full.py:21-56)"""
    model.train()
    train_loader.sampler.set_epoch(epoch)
    acc_grad = AccGrad(get_params(model, ent))
    sim_func = SSIM(11, 3).to(device)
    gamma, beta, alpha, clip = args.gamma, args.beta, args.alpha, args.clip
    log.log('clip:{}'.format(clip))
    acc_batch = args.acc_batch
    last = None
    for batch_idx, data in enumerate(train_loader):
        if not data.shape[0] == args.batch_size:
            continue
        if args.max_steps and batch_idx >= args.max_steps:
            break
        if getattr(args, 'deadline', None) and time.monotonic() >= args.deadline:
            break
        data = data.to(device)
        optimizer.zero_grad()
        optimizer_quant.zero_grad()
        mse_loss, ssim, ent_loss = forward_losses(args, model, data, pr1, pr2, sim_func)
        ssim_loss = 1 - ssim
        loss = gamma * mse_loss + beta * ssim_loss
        if ent_loss is not None:
            loss = loss + alpha * ent_loss
        loss.backward()
        optimizer_quant.step()
        param = list(get_params(model, ent))
        if batch_idx % acc_batch == acc_batch - 1:
            acc_grad.copy_back(param)
            if clip > 0:
                torch.nn.utils.clip_grad_norm_(param, clip)
            optimizer.step()
        else:
            acc_grad.acc(param)
        last = (loss.item(), mse_loss.item(), 1 - ssim_loss.item(), float('nan') if ent_loss is None else ent_loss.item())
        log.log('Train Epoch: {} [{}/{} ({:.0f}%)]\tLoss: {:.6f} mse:{:.6f} ssim:{:.3} rate:{:.3}'.format(
            epoch, batch_idx * len(data), len(train_loader.dataset), 100. * batch_idx / len(train_loader), *last))
    return last


def test(args, model, device, test_loader, log, pr1, pr2):
    """viewport MSE / SSIM / rate over the test set, scored against the anchor curve
    (reference: trainDDP_Full.py:58-86)"""
    model.eval()
    sim_func = SSIM(11, 3).to(device)
    test_mse, test_ssim, test_ent, n = 0., 0., 0., 0
    vd = args.valid_dim / 256. * .815
    for data in test_loader:
        with torch.no_grad():
            mse, ssim, rate = forward_losses(args, model, data.to(device), pr1, pr2, sim_func)
        test_mse += mse.item()
        test_ssim += ssim.item()
        test_ent += 0. if rate is None else rate.item()
        n += 1
    test_mse, test_ssim, test_ent = test_mse / max(n, 1), test_ssim / max(n, 1), test_ent / max(n, 1)
    if args.base:
        log.log('\nTest set: MSE loss: {:.6f}  ssim loss: {:.4f}'.format(test_mse, test_ssim))
        rt_loss = [args.gamma * test_mse + args.beta * (1 - test_ssim)]
    else:
        real_rt = vd * test_ent / 0.693
        log.log('\nTest set: MSE loss: {:.6f}  ssim loss: {:.4f} Ent: {:.3f} rt: {:.3f}bpp'.format(
            test_mse, test_ssim, test_ent, real_rt))
        rt_loss = [float(test_mse - mse_tb(real_rt))]
    log.log(('tloss: ' + '{}\t' * len(rt_loss)).format(*rt_loss))
    return rt_loss


def init_with_trained_model(path, model, device):
    """copy every tensor of the checkpoint whose name exists in the model (reference: :93-100)"""
    pdict = torch.load(path, map_location=device)
    ndict = model.state_dict()
    for key in ndict.keys():
        if key in pdict.keys():
            ndict[key] = pdict[key]
    model.load_state_dict(ndict)


def setup(rank, world_size, backend):
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '12355')
    dist.init_process_group(backend=backend, rank=rank, world_size=world_size)


def make_datasets(args):
    if args.procedural:
        ntest = max(args.test_batch_size, min(8, args.procedural // 8))
        train_data = ProceduralSphereDataSet(args.procedural, args.height, args.width, seed=args.seed * 2 + 1)
        test_data = ProceduralSphereDataSet(ntest, args.height, args.width, seed=args.seed * 2 + 2)
        return train_data, test_data, train_data.values()
    if args.synthetic:
        ntest = max(args.test_batch_size, args.synthetic // 8)
        train_data = SyntheticSphereDataSet(args.synthetic, args.height, args.width, seed=1)
        test_data = SyntheticSphereDataSet(ntest, args.height, args.width, seed=2)
        return train_data, test_data, train_data.values()
    train_data = SphereDataSet(True, args.data_dir, args.train_list)
    test_data = SphereDataSet(False, args.data_dir, args.test_list)
    return train_data, test_data, args.values


def Job(rank, world_size, args):
    """what one rank does (reference: trainDDP_Full.py:102-163)"""
    on_gpu = args.device == 'cuda'
    cid = int(os.environ.get('LOCAL_RANK', rank)) if (on_gpu and not args.share_gpu) else 0
    args.gpu_id = cid
    torch.manual_seed(args.seed)  # same initial weights on every rank; DDP broadcasts rank 0's anyway
    setup(rank, world_size, args.dist_backend or ('nccl' if on_gpu else 'gloo'))
    device = torch.device('cuda:%d' % cid) if on_gpu else torch.device('cpu')
    if on_gpu:
        torch.cuda.set_device(device)
    train_data, test_data, values = make_datasets(args)
    train_loader, test_loader = load_train_test_distribute(
        world_size, rank, args.batch_size, args.test_batch_size, mean=args.mean, acc_batch=args.acc_batch,
        train_data=train_data, test_data=test_data, values=values, num_workers=args.workers)
    save_dir = os.path.join(args.base_dir, 'save_models')
    if rank == 0:
        os.makedirs(save_dir, exist_ok=True)
    dist.barrier()
    prex = '{}_{}_{}_{}_{}'.format('base' if args.base else 'ent', 'opt' if args.opt else 'normal', args.channels,
                                   args.valid_dim, args.npart)
    prex = '{}_init'.format(prex) if args.init else prex
    log = Logger('{}/{:s}_logs_{}.txt'.format(save_dir, prex, cid), screen=args.verbose and rank == 0, file=(rank == 0))
    vs = args.viewport_size
    pr1 = MultiProject(vs, int(vs * 1.5), 0.5, False, cid).to(device)
    pr2 = MultiProject(vs, int(vs * 1.5), 0.5, False, cid).to(device)
    net_class = model_zoo_v2.CMPNetV2M if args.base else model_zoo_v2.CMPNetV2MF
    model = net_class(args.valid_dim, args.channels, args.code_dim, args.npart, opt=args.opt, init=args.init,
                      device_id=cid)
    saver = ModuleSaver(save_dir + '/', prex) if rank == 0 else None
    wrap = lambda m: DDP(m.to(device), [cid] if on_gpu else None)
    best, latest = '{}/{}_best_0.pt'.format(save_dir, prex), '{}/{}_latest.pt'.format(save_dir, prex)
    if args.init:
        # stage 2 trains the entropy model on top of the stage-1 transforms: an own checkpoint of this
        # stage, else --init-from, else what `--base` wrote (trainDDP_Full.py:118-122 loads
        # save_models/base_opt_192_{valid_dim}_16_best_0.pt).  Never from random transforms.
        base_best = '{}/base_{}_{}_{}_{}_best_0.pt'.format(save_dir, 'opt' if args.opt else 'normal', args.channels,
                                                           args.valid_dim, args.npart)
        of = best if os.path.exists(best) else (args.init_from or base_best)
        if not os.path.exists(of) and not args.init_random:
            raise FileNotFoundError(
                '--init needs the stage-1 transforms: no checkpoint at {} (run --base first or pass --init-from; '
                '--init-random trains on random transforms, for smoke runs only)'.format(of))
        if os.path.exists(of):
            init_with_trained_model(of, model, device)
            log.log('load init model {} successful...'.format(of))
        else:
            log.log('WARNING: --init-random: no stage-1 checkpoint at {}, the entropy model is trained on RANDOM transforms'.format(of))
        model = wrap(model)
    elif os.path.exists(best):
        of = latest if (args.latest and os.path.exists(latest)) else best
        init_with_trained_model(of, model, device)
        model = wrap(model)
        ls = test(args, model, device, test_loader, log, pr1, pr2)
        if args.restart:
            ls = [1e9 for _ in range(len(ls))]
        if rank == 0:
            saver.init_loss(ls)
        log.log('load model successful...')
    else:
        of = args.init_from or '{}/{}_init_best_0.pt'.format(save_dir, prex)
        if os.path.exists(of):
            init_with_trained_model(of, model, device)
            log.log('initialize the model with {}...'.format(of))
        else:
            log.log('no checkpoint at {}: training from random weights'.format(of))
        model = wrap(model)
    optimizer_quant = torch.optim.SGD([model.module.quant.count], lr=0.001)
    optimizer_other = torch.optim.Adam([{'params': model.module.encoder.parameters()},
                                        {'params': model.module.decoder.parameters()},
                                        {'params': [model.module.quant.weight]}], lr=args.lr)
    optimizer_ent = None if args.base else torch.optim.Adam(model.module.ent.parameters(), lr=args.lr * 10)
    log.log('lr:{}'.format(args.lr))
    log.log('valid dims:{} \t alpha:{}'.format(args.valid_dim, args.alpha))
    history = []
    args.deadline = time.monotonic() + args.time_budget if args.time_budget > 0 else None
    for epoch in range(1, args.epochs + 1):
        if args.deadline and time.monotonic() >= args.deadline:
            log.log('time budget of {} s used up after {} epoch(s)'.format(args.time_budget, epoch - 1))
            break
        if args.base or (not args.init and epoch % 4 == 1):
            last = train(args, model, device, train_loader, optimizer_other, optimizer_quant, epoch, log, pr1, pr2, False)
        else:
            last = train(args, model, device, train_loader, optimizer_ent, optimizer_quant, epoch, log, pr1, pr2, True)
        ls = test(args, model, device, test_loader, log, pr1, pr2)
        history.append((last, ls))
        if rank == 0:
            log.log(saver.save(model, ls))
    if world_size > 1:
        # the ranks end with the same parameters (DDP averaged every gradient): spread of a checksum over the ranks
        chk = torch.zeros(1, dtype=torch.float64, device=device)
        for p in model.module.parameters():
            chk += p.detach().double().abs().sum()
        hi, lo = chk.clone(), chk.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        log.log('parameter checksum over {} ranks: {:.9e}, spread {:.3e}'.format(world_size, hi.item(), (hi - lo).item()))
    dist.barrier()
    dist.destroy_process_group()
    return history


def build_parser():
    parser = argparse.ArgumentParser(description='PyTorch 360 Compression (MI355X)')
    parser.add_argument('--gpus', type=int, default=0,
                        help='start this many ranks on this node (0: ranks come from the launcher environment)')
    parser.add_argument('--device', default='cuda', choices=['cuda', 'cpu'],
                        help='cpu = gloo ranks on the test backend (tests only)')
    parser.add_argument('--batch-size', type=int, default=4, metavar='N')
    parser.add_argument('--acc-batch', type=int, default=3)
    parser.add_argument('--test-batch-size', type=int, default=4, metavar='N')
    parser.add_argument('--epochs', type=int, default=30, metavar='N')
    parser.add_argument('--lr', type=float, default=0.0001, metavar='LR')
    parser.add_argument('--valid-dim', type=int, default=192)
    parser.add_argument('--gamma', type=float, default=1, help='trade-off of MSE loss')
    parser.add_argument('--beta', type=float, default=0, help='trade-off of SSIM loss')
    parser.add_argument('--alpha', type=float, default=1, help='trade-off of rate loss')
    parser.add_argument('--clip', type=float, default=0.1,
                        help='global gradient-norm clip per optimiser step; 0 = none.  (The reference passes an exhausted '
                             'generator to clip_grad_norm_, trainDDP_Full.py:43-46, so its runs are in effect unclipped: '
                             'use --clip 0 to reproduce them)')
    parser.add_argument('--opt', action='store_true', default=True, help='optimised tile split')
    parser.add_argument('--no-opt', dest='opt', action='store_false')
    parser.add_argument('--init', action='store_true', default=False,
                        help='first stage: train the entropy model only, its gradient cut off from the codes')
    parser.add_argument('--base', action='store_true', default=False,
                        help='the transforms only, no entropy model (trainDDP_Base.py)')
    parser.add_argument('--latest', action='store_true', default=False)
    parser.add_argument('--restart', action='store_true', default=False)
    parser.add_argument('--viewport_size', type=int, default=171, metavar='viewport')
    parser.add_argument('--channels', type=int, default=192)
    parser.add_argument('--code-dim', type=int, default=192)
    parser.add_argument('--npart', type=int, default=16)
    parser.add_argument('--mean', type=float, default=1.5, help="the sampler's minimum mean image value per step")
    parser.add_argument('--base-dir', default='.', help='checkpoints and logs go to <base-dir>/save_models')
    parser.add_argument('--init-from', default=None, help='checkpoint to initialise from')
    parser.add_argument('--init-random', action='store_true', default=False,
                        help='--init without a stage-1 checkpoint: go on from random transforms (smoke runs only)')
    parser.add_argument('--data-dir', default='./360_512')
    parser.add_argument('--train-list', default=None)
    parser.add_argument('--test-list', default=None)
    parser.add_argument('--values', default=None, help='pickle: image name -> value, for the balanced sampler')
    parser.add_argument('--synthetic', type=int, default=0, help='train on this many generated images')
    parser.add_argument('--procedural', type=int, default=0,
                        help='train on this many procedural images (gradients, textures, hard-edged shapes)')
    parser.add_argument('--time-budget', type=float, default=0,
                        help='stop after this many seconds of wall time (0: run all epochs)')
    parser.add_argument('--height', type=int, default=512)
    parser.add_argument('--width', type=int, default=1024)
    parser.add_argument('--workers', type=int, default=4)
    parser.add_argument('--max-steps', type=int, default=0, help='stop an epoch after this many batches (0: all)')
    parser.add_argument('--dist-backend', default=None, choices=['nccl', 'gloo'],
                        help='default: nccl (RCCL) on the GPU, gloo on the CPU.  gloo with --device cuda = several ranks on ONE '
                             'GPU (RCCL refuses two ranks per device): a rehearsal of the DDP path, not a deployment')
    parser.add_argument('--share-gpu', action='store_true', default=False,
                        help='every rank on cuda:0 (with --dist-backend gloo): rehearsal on a one-GPU box')
    parser.add_argument('--seed', type=int, default=0)
    parser.add_argument('--verbose', action='store_true', default=False)
    return parser


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = build_parser().parse_args(argv)
    if args.gpus > 0 and 'WORLD_SIZE' not in os.environ:
        # become the launcher: one child per GPU through torch.distributed.run
        import subprocess
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', os.environ.get('MASTER_PORT', '29517'),
               '-m', 'pseudocylindrical_convolution_amd.train'] + argv
        return subprocess.call(cmd)
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    Job(rank, world, args)
    return 0


if __name__ == '__main__':
    sys.exit(main())
