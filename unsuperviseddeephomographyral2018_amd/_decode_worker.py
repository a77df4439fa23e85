"""Image-decode worker process of dataloader.Dataloader(num_workers=K).

The reference decodes inside TF queue runners -- native threads (dataloader.py:230-246).  Python threads share one
interpreter lock around PIL's non-decoding work, which caps a thread pool at ~4 images/ms whatever the core count, so
the loader can start K of these instead: plain `python -m ..._decode_worker <frames file> <slots> <img_h> <img_w>`
children (they import numpy + PIL only, never torch / HIP) that say "ready" once the ring is mapped (the parent then
unlinks the file: nothing is left behind whatever happens to either side), read "slot<space>path" lines on stdin, decode the file
(area-resize to img_h x img_w if needed, as `dataloader._decode`) straight into slot `slot` of a uint8
[slots, img_h, img_w, 3] array memory-mapped from a file under /dev/shm, and answer "slot" (or "slot !message") on stdout.
"""
import sys

import numpy as np


def main(argv):
    path, slots, img_h, img_w = argv[0], int(argv[1]), int(argv[2]), int(argv[3])
    from PIL import Image
    frames = np.memmap(path, dtype=np.uint8, mode='r+', shape=(slots, img_h, img_w, 3))
    out = sys.stdout
    out.write('ready\n')                          # the ring is mapped: the parent may unlink the file now
    out.flush()
    for line in sys.stdin:
        line = line.rstrip('\n')
        if not line:
            continue
        slot, name = line.split(' ', 1)
        try:
            with Image.open(name) as im:
                im = im.convert('RGB')
                if im.size != (img_w, img_h):
                    im = im.resize((img_w, img_h), Image.BOX)
                frames[int(slot)] = np.asarray(im, dtype=np.uint8)
            out.write(slot + '\n')
        except Exception as e:                      # reported to the parent, which raises
            out.write('%s !%s: %s\n' % (slot, type(e).__name__, str(e).replace('\n', ' ')[:300]))
        out.flush()


if __name__ == '__main__':
    main(sys.argv[1:])
