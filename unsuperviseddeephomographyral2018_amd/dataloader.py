"""Dataloader -- host-side mirror of /root/reference/code/dataloader.py for the reference's ON-DISK format.

The reference's loader is a TF1 queue-runner graph: read `filenames_file` / `pts1_file` / `gt_file`
(dataloader.py:49-72), decode `I/<name>` and `I_prime/<name>` (JPEG/PNG, :236-246), augment, standardise, gray
patches, patch indices (:160-227), `shuffle_batch` with 20 threads (:230-235).  Here the per-sample arithmetic is
ONE HIP kernel over a batch of decoded uint8 frames in HBM (csrc/uh_inputs.hip, `uh_prepare_inputs`); the host only
parses the text files, decodes images (PIL, a thread pool -- JPEG entropy decoding stays on the CPU) and draws the
augmentation parameters.  Same namedtuple, same text formats (np.savetxt rows, "a.jpg b.jpg" lines:
utils/gen_synthetic_data.py:121-126), same output contract: the 9 tensors HomographyModel takes, NHWC f32.

`Dataloader(num_workers=K)` decodes in K worker PROCESSES (`_decode_worker.py`: numpy + PIL only) into a shared uint8
frame ring under /dev/shm instead of the thread pool -- Python threads serialise on the interpreter lock at ~2 000
pairs/s, which is a fifth of what the train step consumes (profiles/r03_inputs_timing.txt).

Differences, stated: shuffling is a seeded permutation per epoch instead of TF's min_after_dequeue window;
`per_image_normalize` is not implemented (the reference's own branch references undefined names, :180-187);
frames whose decoded size differs from (img_h, img_w) are area-resized with PIL's BOX filter (TF's AREA resize).
"""
import ctypes as C
import os
from collections import namedtuple
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import _lib

# dataloader.py:11-22
dataloader_params = namedtuple('parameters',
                               'data_path,'
                               'filenames_file,'
                               'pts1_file,'
                               'gt_file,'
                               'mode,'
                               'batch_size,'
                               'img_h,'
                               'img_w,'
                               'patch_size,'
                               'augment_list,'
                               'do_augment,')

MEAN_I = (118.93, 113.97, 102.60)          # dataloader.py:99
STD_I = (69.85, 68.81, 72.45)              # dataloader.py:100


def read_img_and_gt(filenames_file, pts1_file, gt_file):
    """dataloader.py:49-72 -> (list of [nameA, nameB], pts1 [N,8] f64, gt [N,8] f64 or None)."""
    with open(pts1_file) as f:
        pts1 = np.array([l.split() for l in (x.strip() for x in f) if l]).astype('float64')
    with open(filenames_file) as f:
        names = [l.split() for l in (x.strip() for x in f) if l]
    if not gt_file:
        return names, pts1, None
    with open(gt_file) as f:
        gt = np.array([l.split() for l in (x.strip() for x in f) if l]).astype('float64')
    return names, pts1, gt


def sample_augmentation(B, mode, do_augment, generator=None):
    """Augmentation parameters [B,2,5] (gamma, brightness, colour r,g,b for I and I') on the CPU.

    Law of dataloader.py:160-166,323-375: with probability `do_augment` a pair is augmented -- gamma ~ U(0.8,1.2),
    brightness ~ U(0.5,2.0), colour ~ U(0.8,1.2)^3 -- JOINTLY (same draw for both images) in training, DISJOINTLY in
    test mode; otherwise the identity (1,1,1,1,1), which the kernel maps to the un-augmented values exactly."""
    g = generator
    u = lambda lo, hi, *shape: lo + (hi - lo) * torch.rand(*shape, generator=g)
    p = torch.cat([u(0.8, 1.2, B, 2, 1), u(0.5, 2.0, B, 2, 1), u(0.8, 1.2, B, 2, 3)], 2)
    if mode == 'train':
        p[:, 1] = p[:, 0]
    apply = torch.rand(B, generator=g) > (1.0 - do_augment)
    ident = torch.ones(B, 2, 5)
    return torch.where(apply[:, None, None], p, ident).contiguous()


def prepare_inputs(I_u8, I_prime_u8, pts1, patch_size, aug=None, mean=MEAN_I, std=STD_I):
    """uint8 frames [B,H,W,3] (on the HIP device) -> dict of the model's input tensors via uh_prepare_inputs."""
    lib = _lib.load()
    for t, n in ((I_u8, 'I_u8'), (I_prime_u8, 'I_prime_u8')):
        if not t.is_cuda or t.dtype != torch.uint8 or t.dim() != 4 or t.shape[3] != 3:
            raise _lib.UHError('%s must be a uint8 [B,H,W,3] tensor on the HIP device' % n)
    I_u8 = I_u8.contiguous(); I_prime_u8 = I_prime_u8.contiguous()
    B, H, W, _ = I_u8.shape
    P = int(patch_size)
    dev = I_u8.device
    pts1 = pts1.to(device=dev, dtype=torch.float32).contiguous()
    if aug is not None:
        aug = aug.to(device=dev, dtype=torch.float32).contiguous()
        if tuple(aug.shape) != (B, 2, 5):
            raise _lib.UHError('aug must be [B,2,5]')
    f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    # the patch tensors are zero-filled: the kernel writes a patch entry only where the patch lies inside the frame
    # (gen_synthetic_data.py:42-53 guarantees it does); a patch sticking out leaves zeros / index 0, never garbage
    z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
    out = dict(I_aug=f(B, H, W, 3), I_prime_aug=f(B, H, W, 3), I1=z(B, P, P, 1), I2=z(B, P, P, 1),
               I1_aug=z(B, P, P, 1), I2_aug=z(B, P, P, 1),
               patch_indices=torch.zeros((B, P * P), dtype=torch.int32, device=dev), pts1=pts1)
    m = (C.c_float * 3)(*mean); s = (C.c_float * 3)(*std)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    _lib.check(lib.uh_prepare_inputs(p(I_u8), p(I_prime_u8), p(aug), p(pts1), m, s, p(out['I_aug']), p(out['I_prime_aug']),
                                     p(out['I1']), p(out['I2']), p(out['I1_aug']), p(out['I2_aug']),
                                     p(out['patch_indices']), B, H, W, P,
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'uh_prepare_inputs')
    return out


def _decode(path, img_h, img_w):
    from PIL import Image
    with Image.open(path) as im:
        im = im.convert('RGB')
        if im.size != (img_w, img_h):
            im = im.resize((img_w, img_h), Image.BOX)
        return np.asarray(im, dtype=np.uint8)


class Dataloader(object):
    """Iterate batches of a dataset stored in the reference's format:
        <data_path>/I/<name>, <data_path>/I_prime/<name>, filenames_file ("a.jpg b.jpg" per line; the second token
        names both files, dataloader.py:148-149), pts1_file, gt_file (one np.savetxt row of 8 floats per pair)."""

    def __init__(self, params, shuffle=True, device='cuda', seed=0, num_threads=20, num_workers=0):
        self.params = params
        self.shuffle = shuffle
        self.device = torch.device(device)
        if self.device.type == 'cuda' and self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.num_workers = int(num_workers)         # > 0: decode in that many worker processes (stream() only)
        self.names, self.pts1, self.gt = read_img_and_gt(params.filenames_file, params.pts1_file, params.gt_file)
        if len(self.names) != len(self.pts1) or (self.gt is not None and len(self.gt) != len(self.names)):
            raise ValueError('filenames / pts1 / gt files disagree on the number of pairs')
        # two streams of randomness: the epoch permutations and the augmentation draws.  (One shared generator would tie
        # the batches' contents to HOW FAR AHEAD the index list is consumed -- the worker-process route reads ahead.)
        self.gen = torch.Generator().manual_seed(seed)
        self.gen_order = torch.Generator().manual_seed(seed + 0x5EED)
        self.pool = ThreadPoolExecutor(max_workers=num_threads)
        if 'per_image_normalize' in params.augment_list:
            raise NotImplementedError('per_image_normalize (the reference branch itself is broken, dataloader.py:180-187)')

    def __len__(self):
        """Batches per epoch: every pair is delivered at least once (the tail batch is completed by wrapping around)."""
        return -(-len(self.names) // self.params.batch_size)

    def _load(self, i):
        pr = self.params
        name = self.names[i][1] if len(self.names[i]) > 1 else self.names[i][0]
        return (_decode(os.path.join(pr.data_path, 'I', name), pr.img_h, pr.img_w),
                _decode(os.path.join(pr.data_path, 'I_prime', name), pr.img_h, pr.img_w))

    def _order(self):
        n = len(self.names)
        return torch.randperm(n, generator=self.gen_order).tolist() if self.shuffle else list(range(n))

    def _batch(self, ids):
        frames = list(self.pool.map(self._load, ids))
        I8 = torch.from_numpy(np.stack([f[0] for f in frames])).to(self.device, non_blocking=True)
        Ip8 = torch.from_numpy(np.stack([f[1] for f in frames])).to(self.device, non_blocking=True)
        return self._finish(I8, Ip8, ids)

    def _finish(self, I8, Ip8, ids):
        """decoded uint8 frames on the device -> the model's input tensors (uh_prepare_inputs)"""
        pr = self.params
        pts1 = torch.from_numpy(self.pts1[ids].astype(np.float32))
        aug = sample_augmentation(len(ids), pr.mode, pr.do_augment, self.gen) if pr.do_augment > 0 else None
        normalize = 'normalize' in pr.augment_list
        batch = prepare_inputs(I8, Ip8, pts1, pr.patch_size, aug,
                               MEAN_I if normalize else (0., 0., 0.), STD_I if normalize else (1., 1., 1.))
        batch['gt'] = (torch.from_numpy(self.gt[ids].astype(np.float32)).to(self.device)
                       if self.gt is not None else torch.zeros(len(ids), 8, device=self.device))
        return batch

    def stream(self, prefetch=0):
        """Endless stream of full batches, the reference's tf.train.(shuffle_)batch queue (dataloader.py:255-262): the
        index list is cycled epoch after epoch (re-shuffled each time), so NO pair is ever dropped and a batch may
        straddle two epochs.  Raises instead of spinning when there is nothing to deliver.
        prefetch > 0: a producer thread keeps up to `prefetch` finished batches queued (the reference's queue runners,
        :230-235), so that image decoding on the host overlaps the training step on the GPU; same batches, same order."""
        B = self.params.batch_size
        if len(self.names) == 0:
            raise ValueError('Dataloader: %s lists no pairs' % self.params.filenames_file)
        if B <= 0:
            raise ValueError('Dataloader: batch_size must be positive')
        gen = self._worker_stream(max(int(prefetch), 1) + 1) if self.num_workers > 0 else self._sync_stream()
        if prefetch > 0:
            return self._prefetched(gen, int(prefetch))
        return gen

    def _id_batches(self):
        B = self.params.batch_size
        pending = []
        while True:
            while len(pending) < B:
                pending.extend(self._order())
            ids, pending = pending[:B], pending[B:]
            yield ids

    def _sync_stream(self):
        for ids in self._id_batches():
            yield self._batch(ids)

    def _worker_stream(self, nbuf):
        """Same batches, same order as _sync_stream, decoded by `num_workers` processes: up to `nbuf` batches are in flight
        in the workers while the oldest one is uploaded and turned into the model's inputs."""
        import subprocess
        import sys
        import tempfile
        pr = self.params
        B, H, W = pr.batch_size, pr.img_h, pr.img_w
        per = 2 * B                                             # frames of one batch: I then I'
        slots = nbuf * per
        shm_dir = None                                          # frame ring: a tmpfs file if /dev/shm has the room, else $TMPDIR
        if os.path.isdir('/dev/shm'):
            fs = os.statvfs('/dev/shm')
            if fs.f_bavail * fs.f_frsize > 2 * slots * H * W * 3:   # (a container's default /dev/shm is 64 MB: a write past
                shm_dir = '/dev/shm'                                #  its capacity would kill the worker with SIGBUS)
        fd, path = tempfile.mkstemp(prefix='uh_frames_', dir=shm_dir)
        os.ftruncate(fd, slots * H * W * 3)
        os.close(fd)
        frames = np.memmap(path, dtype=np.uint8, mode='r+', shape=(nbuf, 2, B, H, W, 3))
        # page-lock the ring so that the uploads are DMA copies that run beside the training step (a pageable source is
        # staged through a bounce buffer by the calling thread); a refusal only means slower, synchronous uploads
        pinned = False
        if self.device.type == 'cuda':
            try:
                pinned = int(torch.cuda.cudart().cudaHostRegister(frames.ctypes.data, frames.nbytes, 0)) == 0
            except Exception:
                pinned = False
        script = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_decode_worker.py')
        K = self.num_workers
        import queue
        import threading
        procs, answers, path_gone = [], [], False

        def unlink_ring():
            nonlocal path_gone
            if not path_gone:
                path_gone = True
                try:
                    os.unlink(path)
                except OSError:
                    pass
        try:
            procs = [subprocess.Popen([sys.executable, script, path, str(slots), str(H), str(W)], stdin=subprocess.PIPE,
                                      stdout=subprocess.PIPE, text=True, bufsize=1) for _ in range(K)]
            # One reader thread per worker drains its stdout into a queue: the parent writes the paths of several batches
            # before it reads an answer, and a worker whose 64 KB stdout pipe is full (long error lines, few workers, big
            # batches) would otherwise block while the parent blocks on ITS stdin -- a deadlock instead of the RuntimeError.
            answers = [queue.Queue() for _ in range(K)]

            def drain(w):
                try:
                    for line in procs[w].stdout:
                        answers[w].put(line)
                finally:
                    answers[w].put('')                           # EOF marker: the worker is gone

            for w in range(K):
                threading.Thread(target=drain, args=(w,), daemon=True).start()

            def answer(w, timeout=None):
                try:
                    ans = answers[w].get(timeout=timeout)
                except queue.Empty:
                    raise RuntimeError('decode worker %d did not answer within %s s' % (w, timeout))
                if not ans:
                    raise RuntimeError('decode worker %d exited (code %s)' % (w, procs[w].poll()))
                return ans.rstrip('\n')
            # every worker reports once it has mapped the ring; the FILE is then unlinked at once -- the mappings (ours, page-
            # locked, and the workers') keep the memory alive, and nothing is left under /dev/shm when this process exits,
            # is interrupted or is killed (a ring is 2 * nbuf * B * H * W * 3 bytes of tmpfs RAM: 147 MB at B=64, 240x320)
            for w in range(K):
                if answer(w, timeout=120) != 'ready':
                    raise RuntimeError('decode worker %d: unexpected greeting' % w)
            unlink_ring()
            ids_iter = self._id_batches()
            inflight = []                                        # (buffer index, ids, frames handed to each worker)
            free = list(range(nbuf))
            rr = 0

            def submit():
                nonlocal rr
                buf = free.pop()
                ids = next(ids_iter)
                count = [0] * K
                for which, sub in ((0, 'I'), (1, 'I_prime')):
                    for j, i in enumerate(ids):
                        name = self.names[i][1] if len(self.names[i]) > 1 else self.names[i][0]
                        w = rr % K; rr += 1
                        procs[w].stdin.write('%d %s\n' % (buf * per + which * B + j, os.path.join(pr.data_path, sub, name)))
                        count[w] += 1
                for p in procs:
                    p.stdin.flush()
                inflight.append((buf, ids, count))

            while True:
                while free:
                    submit()
                buf, ids, count = inflight.pop(0)
                for w, n in enumerate(count):                    # every worker answers its lines in order
                    for _ in range(n):
                        ans = answer(w)
                        body = ans.split(' ', 1)
                        if len(body) > 1 and body[1].startswith('!'):          # "slot !ErrorType: message"
                            raise RuntimeError('decode worker: ' + ans)
                I8 = torch.from_numpy(frames[buf, 0]).to(self.device, non_blocking=pinned)
                Ip8 = torch.from_numpy(frames[buf, 1]).to(self.device, non_blocking=pinned)
                if pinned:                                       # the ring slot is free again once the DMA has read it
                    torch.cuda.current_stream(self.device).synchronize()
                # (pageable source: the copy has read the ring buffer when .to() returns)
                free.append(buf)
                yield self._finish(I8, Ip8, ids)
        finally:
            for p in procs:
                try:
                    p.stdin.close()
                except Exception:
                    pass
            for p in procs:
                try:
                    p.wait(timeout=5)
                except Exception:
                    p.kill()
            if pinned:
                try:
                    torch.cuda.cudart().cudaHostUnregister(frames.ctypes.data)
                except Exception:
                    pass
            del frames
            unlink_ring()

    def _prefetched(self, gen, depth):
        import queue
        import threading
        q = queue.Queue(maxsize=depth)
        stop = threading.Event()

        def producer():
            try:
                side = None
                if self.device.type == 'cuda':
                    torch.cuda.set_device(self.device)
                    # uploads and uh_prepare_inputs run on a stream of their own, so that they overlap the training step
                    # instead of queueing behind it on the default stream
                    side = torch.cuda.Stream(device=self.device)
                    torch.cuda.set_stream(side)               # (thread-local: only this producer thread)
                for batch in gen:
                    ev = None
                    if side is not None:                      # the consumer's stream waits for the producer's kernels
                        ev = torch.cuda.Event()
                        ev.record(side)
                    while not stop.is_set():
                        try:
                            q.put((batch, ev), timeout=0.1)
                            break
                        except queue.Full:
                            continue
                    if stop.is_set():
                        return
            except BaseException as e:                        # delivered to the consumer, not swallowed
                q.put((e, None))
            finally:
                gen.close()                                   # stops the decode workers / frees the frame ring, if any

        worker = threading.Thread(target=producer, daemon=True)
        worker.start()
        try:
            while True:
                item, ev = q.get()
                if isinstance(item, BaseException):
                    raise item
                if ev is not None:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ev)
                    for ten in item.values():                 # allocated on the producer's stream, consumed on this one
                        if torch.is_tensor(ten) and ten.is_cuda:
                            ten.record_stream(cur)
                yield item
        finally:
            # closing (or dropping) this generator must END the producer: it is the producer's `finally` that stops the decode
            # workers and releases the page-locked frame ring; a daemon thread that is merely abandoned never gets there
            stop.set()
            try:
                while True:                                   # a producer blocked on a full queue sees `stop` within 0.1 s;
                    q.get_nowait()                            # draining lets one blocked on put() of an exception return too
            except queue.Empty:
                pass
            worker.join(timeout=30)

    def __iter__(self):
        """One epoch = ceil(n / batch_size) full batches; the last one is completed with the first pairs of the next
        pass, so every pair is seen (the round-1 loader dropped the remainder and yielded nothing for n < batch_size)."""
        it = self.stream()
        for _ in range(len(self)):
            yield next(it)


def write_dataset(data_path, I_u8, I_prime_u8, pts1, gt, prefix='', fmt='png'):
    """Write pairs in the reference's on-disk layout (utils/gen_synthetic_data.py:100-126): I/<i>.<fmt>,
    I_prime/<i>.<fmt>, <prefix>filenames.txt, <prefix>pts1.txt, <prefix>gt.txt.  Returns the three text paths."""
    from PIL import Image
    os.makedirs(os.path.join(data_path, 'I'), exist_ok=True)
    os.makedirs(os.path.join(data_path, 'I_prime'), exist_ok=True)
    ff = os.path.join(data_path, prefix + 'filenames.txt')
    fp = os.path.join(data_path, prefix + 'pts1.txt')
    fg = os.path.join(data_path, prefix + 'gt.txt')
    with open(ff, 'w') as f_names, open(fp, 'w') as f_pts1, open(fg, 'w') as f_gt:
        for i in range(len(I_u8)):
            name = '%d.%s' % (i, fmt)
            Image.fromarray(np.asarray(I_u8[i], np.uint8)).save(os.path.join(data_path, 'I', name))
            Image.fromarray(np.asarray(I_prime_u8[i], np.uint8)).save(os.path.join(data_path, 'I_prime', name))
            np.savetxt(f_gt, [np.asarray(gt[i], np.float32)], delimiter=' ')
            np.savetxt(f_pts1, [np.asarray(pts1[i], np.float32)], delimiter=' ')
            f_names.write('%s %s\n' % (name, name))
    return ff, fp, fg
