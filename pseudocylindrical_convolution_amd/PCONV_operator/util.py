"""Small helpers of the reference package: Logger, ModuleSaver, Timer
(reference: PCONV_operator/Logger.py, ModuleSaver.py, Mtimer.py)."""
import os

import torch


class Logger(object):

    def __init__(self, fname, screen=True, file=True):
        self.file = file
        self.fout = open(fname, 'w') if file else None
        self.screen_out = screen

    def log(self, *args):
        if self.screen_out:
            print(*args)
        if self.file:
            self.fout.write(' '.join(str(a) for a in args))
            self.fout.write('\n')
            self.fout.flush()

    def close(self):
        if self.file and self.fout and not self.fout.closed:
            self.fout.close()

    def __del__(self):
        self.close()


class ModuleSaver(object):
    """keeps `<prex>_best_<i>.pt` per tracked loss and `<prex>_latest.pt` otherwise"""

    def __init__(self, path='./saved_models/', prex='default'):
        self.path, self.prex = path, prex
        os.makedirs(path, exist_ok=True)
        self.current_best_loss = None
        self.init = False

    def init_loss(self, loss):
        self.current_best_loss = list(loss) if isinstance(loss, list) else [loss]
        self.init = True

    def save(self, model, loss):
        wrapped = isinstance(model, (torch.nn.DataParallel, torch.nn.parallel.DistributedDataParallel))
        state = model.module.state_dict() if wrapped else model.state_dict()
        loss = loss if isinstance(loss, list) else [loss]
        if not self.init:
            self.current_best_loss = [10e9] * len(loss)
            self.init = True
        notes = []
        for i, value in enumerate(loss):
            if value < self.current_best_loss[i]:
                self.current_best_loss[i] = value
                torch.save(state, os.path.join(self.path, '%s_best_%d.pt' % (self.prex, i)))
                notes.append('save %s_best_%d.pt' % (self.prex, i))
        if not notes:
            torch.save(state, os.path.join(self.path, '%s_latest.pt' % self.prex))
            return 'update %s_latest.pt' % self.prex
        return '\t'.join(notes) + '\t'


class Timer(object):
    """event-pair GPU timer that prints elapsed ms when enabled"""

    def __init__(self, flag=False):
        self.flag = flag
        if flag:
            self.start_t = torch.cuda.Event(enable_timing=True)
            self.end_t = torch.cuda.Event(enable_timing=True)

    def start(self):
        if self.flag:
            self.start_t.record()

    def end(self, out_string=''):
        if self.flag:
            self.end_t.record()
            torch.cuda.synchronize()
            print(out_string, self.start_t.elapsed_time(self.end_t))
