"""ERP image datasets and the rate-balanced distributed sampler of the training scripts
(reference: test/SphereDataset.py).

The reference hard-wires its data paths; here they are arguments.  `SyntheticSphereDataSet`
stands in where no images exist (tests, benchmarks): seeded smooth random ERP images with a
per-image detail level, which also serves as the sampler's per-image value."""
import os
import pickle

import numpy as np
import torch
from torch.utils.data import Dataset


class SphereDataSet(Dataset):
    """images named in a list file, as float32 CxHxW in [0, 1], BGR channel order
    (reference: SphereDataset.py:8-36; `.npy` entries are uint8 HxWx3 arrays)"""

    def __init__(self, train=True, img_dir='./360_512', list_file=None):
        self.img_path = img_dir
        if list_file is None:
            list_file = os.path.join(os.path.dirname(img_dir.rstrip('/')), 'train.txt' if train else 'test.txt')
        with open(list_file) as f:
            self.img_list = [line.rstrip('\n') for line in f if line.strip()]

    def __len__(self):
        return len(self.img_list)

    def __getitem__(self, idx):
        if torch.is_tensor(idx):
            idx = idx.tolist()
        name = os.path.join(self.img_path, self.img_list[idx])
        if name.endswith('.npy'):
            img = np.load(name)
        else:
            from .pseudo_codec import read_image
            img = read_image(name)
        return (torch.from_numpy(img.transpose(2, 0, 1).copy()).type(torch.float32) / 255.0).contiguous()


class SyntheticSphereDataSet(Dataset):
    """`count` seeded random ERP images (height x width): a few low-frequency cosine waves plus noise
    whose amplitude is the image's detail level; `values()` returns those levels"""

    def __init__(self, count, height=512, width=1024, seed=0):
        self.count, self.height, self.width, self.seed = int(count), int(height), int(width), int(seed)
        self.img_list = ['synthetic_%06d.png' % i for i in range(self.count)]
        g = torch.Generator().manual_seed(self.seed)
        self.detail = (0.5 + 2.0 * torch.rand(self.count, generator=g)).tolist()

    def values(self):
        return {name: v for name, v in zip(self.img_list, self.detail)}

    def __len__(self):
        return self.count

    def __getitem__(self, idx):
        if torch.is_tensor(idx):
            idx = idx.tolist()
        g = torch.Generator().manual_seed(self.seed * 1000003 + int(idx))
        yy = torch.linspace(0, 1, self.height).view(1, -1, 1)
        xx = torch.linspace(0, 1, self.width).view(1, 1, -1)
        img = torch.full((3, self.height, self.width), 0.5)
        for _ in range(4):
            fy, fx = (torch.rand(2, generator=g) * 6).tolist()
            phase = torch.rand(3, 1, 1, generator=g) * 6.2832
            img = img + 0.1 * torch.cos(6.2832 * (fy * yy + round(fx) * xx) + phase)
        img = img + 0.02 * self.detail[idx] * torch.randn(3, self.height, self.width, generator=g)
        return img.clamp_(0, 1).contiguous()


class ProceduralSphereDataSet(Dataset):
    """`count` seeded procedural ERP images for training where no photographs exist: a smooth colour
    gradient (low-frequency waves, integer horizontal frequencies so that the frame wraps), a
    band-limited texture (octaves of bilinearly up-sampled noise with a random fall-off), flat or
    shaded shapes with hard edges (rectangles, ellipses, half planes, stripes) and a little sensor
    noise.  The per-image value the balanced sampler reads is the image's detail level."""

    def __init__(self, count, height=512, width=1024, seed=0):
        self.count, self.height, self.width, self.seed = int(count), int(height), int(width), int(seed)
        self.img_list = ['procedural_%06d.png' % i for i in range(self.count)]
        g = torch.Generator().manual_seed(self.seed)
        self.detail = (0.5 + 2.0 * torch.rand(self.count, generator=g)).tolist()

    def values(self):
        return {name: v for name, v in zip(self.img_list, self.detail)}

    def __len__(self):
        return self.count

    def __getitem__(self, idx):
        if torch.is_tensor(idx):
            idx = idx.tolist()
        return procedural_erp(self.height, self.width, self.seed * 1000003 + int(idx), self.detail[int(idx)])


def procedural_erp(height, width, seed, detail=1.5):
    """one (3, height, width) image in [0, 1]; `detail` in [0.5, 2.5] scales texture and shape count"""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(int(seed))
    H, W = int(height), int(width)
    yy = torch.linspace(0, 1, H).view(1, H, 1)
    xx = torch.linspace(0, 1, W).view(1, 1, W)

    def rnd(*shape):
        return torch.rand(*shape, generator=g)

    def texture(channels, octaves, falloff, coarsest):
        t = torch.zeros(channels, H, W)
        amp = 1.0
        for o in range(octaves):
            s = max(1, coarsest >> o)
            hh, ww = max(2, H // s), max(2, W // s)
            n = rnd(1, channels, hh, ww) - 0.5
            n = torch.cat([n, n[..., :1]], 3)                       # the noise wraps with the frame
            t += amp * F.interpolate(n, size=(H, W + W // ww), mode='bilinear', align_corners=False)[0, :, :, :W]
            amp *= falloff
        return t

    # smooth gradient between two colours
    c0, c1 = rnd(3, 1, 1), rnd(3, 1, 1)
    mix = 0.5 * torch.ones(1, H, W)
    for _ in range(3):
        fy, fx = float(rnd(1)) * 3, int(rnd(1) * 3)
        mix = mix + 0.25 * torch.cos(6.2832 * (fy * yy + fx * xx) + 6.2832 * float(rnd(1)))
    img = c0 + (c1 - c0) * mix.clamp(0, 1)
    # band-limited texture, coarse to fine
    img = img + (0.10 + 0.08 * detail) * texture(3, 5, 0.35 + 0.3 * float(rnd(1)), 64)
    # shapes with hard edges
    for _ in range(int(2 + 3 * detail + 4 * float(rnd(1)))):
        kind = int(rnd(1) * 4)
        cy, cx = float(rnd(1)), float(rnd(1))
        ry, rx = 0.02 + 0.25 * float(rnd(1)), 0.01 + 0.15 * float(rnd(1))
        dx = (xx - cx + 0.5) % 1.0 - 0.5                             # horizontal distance on the circle
        dy = yy - cy
        if kind == 0:
            m = (dx.abs() < rx) & (dy.abs() < ry)
        elif kind == 1:
            m = (dx / rx) ** 2 + (dy / ry) ** 2 < 1.0
        elif kind == 2:
            a = 6.2832 * float(rnd(1))
            m = ((dx * np.cos(a) + dy * np.sin(a)) > 0) & (dy.abs() < ry)
        else:
            period = 0.01 + 0.05 * float(rnd(1))
            m = (((dx + 2 * dy) / period).floor() % 2 == 0) & (dx.abs() < rx) & (dy.abs() < ry)
        col = rnd(3, 1, 1)
        shade = 1.0 + 0.3 * (rnd(1) - 0.5) * (dy / ry).clamp(-1, 1)
        img = torch.where(m, (col * shade).expand(3, H, W), img)
    img = img + 0.004 * detail * torch.randn(3, H, W, generator=g)
    return img.clamp_(0, 1).contiguous()


def balance_windows(indices, value_of, window, threshold):
    """Reorder `indices` in place so that every run of `window` consecutive entries (one optimiser
    step over all ranks and accumulation rounds) has sum(value) >= threshold
    (reference: SphereDataset.py:50-92, MyDistributeSampler.check_modify).

    A window below the threshold gives its smallest element to a donor -- the first later-scanned
    window that is above the threshold and owns an element it can lose while staying above it -- and
    receives the largest such element.  Returns False when no donor is left (the caller reshuffles
    with another seed)."""
    nwin = len(indices) // window
    val = lambda pos: value_of(indices[pos])

    def summary(w):
        vals = [val(w * window + j) for j in range(window)]
        j = min(range(window), key=lambda k: (vals[k], k))
        return j, vals[j], sum(vals)

    def donor_element(vals, incoming, total):
        for j in sorted(range(len(vals)), key=lambda k: vals[k], reverse=True):
            if total - vals[j] + incoming > threshold:
                return j
        return -1

    arg_min, v_min, total = [0] * nwin, [0.0] * nwin, [0.0] * nwin
    for w in range(nwin):
        arg_min[w], v_min[w], total[w] = summary(w)
    first_donor, last = 0, -1
    for w in range(nwin):
        while total[w] < threshold:
            while first_donor < nwin and total[first_donor] < threshold + 0.618 and w > 0:
                first_donor += 1
            if first_donor >= nwin:
                return False
            found = None
            for k in range(first_donor, nwin):
                if total[k] > threshold:
                    j = donor_element([val(k * window + i) for i in range(window)], v_min[w], total[k])
                    if j >= 0:
                        found = (k, j)
                        break
            if found is None:
                return False  # (the reference indexes with a stale donor here)
            k, j = found
            if last == k:
                first_donor += 1
            a, b = w * window + arg_min[w], k * window + j
            indices[a], indices[b] = indices[b], indices[a]
            arg_min[w], v_min[w], total[w] = summary(w)
            arg_min[k], v_min[k], total[k] = summary(k)
            last = k
    return True


class MyDistributeSampler(torch.utils.data.distributed.DistributedSampler):
    """DistributedSampler whose global order is re-balanced so that every optimiser step sees images
    of at least `mean` average value (reference: SphereDataset.py:38-115).  `values`: dict file name ->
    value (the reference's `train_val.dic`), a path to its pickle, or None for no balancing."""

    def __init__(self, dataset, num_replicas, rank, batch_size, shuffle=True, seed=0, mean=1.4, acc_batch=1,
                 values=None):
        super(MyDistributeSampler, self).__init__(dataset, num_replicas, rank, shuffle, seed)
        if isinstance(values, str):
            with open(values, 'rb') as f:
                values = pickle.load(f)
        self.vdict = values
        self.flist = [name.replace('npy', 'png') for name in self.dataset.img_list] \
            if self.dataset.img_list and self.dataset.img_list[0].find('npy') >= 0 else list(self.dataset.img_list)
        self.ws = batch_size * num_replicas * acc_batch
        self.thr = mean * self.ws
        self.seed_ext = 0

    def global_order(self):
        """the balanced order over all ranks for the current epoch"""
        while True:
            if self.shuffle:
                g = torch.Generator()
                g.manual_seed(self.seed + self.epoch + self.seed_ext)
                indices = torch.randperm(len(self.dataset), generator=g).tolist()
            else:
                indices = list(range(len(self.dataset)))
            indices += indices[:(self.total_size - len(indices))]
            assert len(indices) == self.total_size
            if self.vdict is None:
                return indices
            if balance_windows(indices, lambda i: self.vdict[self.flist[i]], self.ws, self.thr) or not self.shuffle:
                return indices
            self.seed_ext += 1
            if self.seed_ext > 1000:
                raise RuntimeError("MyDistributeSampler: mean=%g cannot be met by this dataset" % (self.thr / self.ws))

    def __iter__(self):
        indices = self.global_order()[self.rank:self.total_size:self.num_replicas]
        assert len(indices) == self.num_samples
        return iter(indices)


def load_train_test_distribute(world_size, rank, batch_size, test_batch_size, shuffle=True, seed=0, mean=1.4,
                               acc_batch=1, train_data=None, test_data=None, values=None, num_workers=4):
    """(train loader with the balanced distributed sampler, test loader) (reference: SphereDataset.py:118-131)"""
    kwargs = {'num_workers': num_workers, 'pin_memory': torch.cuda.is_available()}
    train_data = train_data if train_data is not None else SphereDataSet(True)
    test_data = test_data if test_data is not None else SphereDataSet(False)
    sampler = MyDistributeSampler(train_data, num_replicas=world_size, rank=rank, batch_size=batch_size,
                                  shuffle=shuffle, seed=seed, mean=mean, acc_batch=acc_batch, values=values)
    train_loader = torch.utils.data.DataLoader(train_data, sampler=sampler, batch_size=batch_size, shuffle=False,
                                               **kwargs)
    test_loader = torch.utils.data.DataLoader(test_data, batch_size=test_batch_size, shuffle=False, **kwargs)
    return train_loader, test_loader
