"""PCIe-inclusive extraction rate: frames in host memory -> feeder (uploads ahead on its own stream) ->
orbx_extract_batch on the consumer stream; and the PNG decode rate of the disk path."""
import io, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, '.')
import torch
from orb_slam_amd import orbx

d = np.load('tests/golden/test_images.npz')
imgs = [d['bgr1'], d['bgr2']]
B, NB = 256, 8
host = torch.from_numpy(np.stack([imgs[i % 2] for i in range(B * NB)])).pin_memory()
frames = host.numpy()
ex = orbx.Extractor(n_key_points=1000, level_pyramid=8, max_width=640, max_height=480, max_batch=B)
stream = torch.cuda.Stream()
ex.set_stream(stream.cuda_stream)
fd = orbx.Feeder(640, 480, 3, B, n_slots=3)

def run(pinned):
    fd.start_memory(frames, already_pinned=pinned)
    n = 0
    while True:
        ptr, k = fd.next(stream.cuda_stream)
        if k == 0:
            break
        ex.extract_batch(ptr, k, 640 * 480 * 3, 640, 480, 640 * 3, 3, 1280, to_host=False)
        n += k
    stream.synchronize()
    return n

for pinned, label in ((True, 'caller-pinned memory'), (False, 'pageable memory (registered or staged by the feeder)')):
    run(pinned)
    t0 = time.perf_counter()
    n = run(pinned)
    dt = time.perf_counter() - t0
    print('%s: %d frames in %.1f ms -> %.0f frames/s (%.1f GB/s over the bus)' %
          (label, n, dt * 1e3, n / dt, n * 640 * 480 * 3 / dt / 1e9))

# disk path: PNG decode rate with the feeder's worker pool
from PIL import Image
tmp = tempfile.mkdtemp() + '/'
os.makedirs(tmp + 'rgb')
N = 128
with open(tmp + 'assoc.txt', 'w') as f:
    for i in range(N):
        Image.fromarray(imgs[i % 2][:, :, ::-1]).save(tmp + 'rgb/%04d.png' % i)
        f.write('%f rgb/%04d.png %f depth/%04d.png\n' % (i / 30, i, i / 30, i))
seq = orbx.Sequence(tmp)
for threads in (1, 16):
    fp = orbx.Feeder(640, 480, 3, 64, n_slots=3, n_threads=threads)
    t0 = time.perf_counter()
    fp.start_sequence(seq, 0, N)
    n = 0
    while True:
        ptr, k = fp.next(None)
        if k == 0:
            break
        n += k
    dt = time.perf_counter() - t0
    print('PNG sequence, %d decode threads: %d frames in %.0f ms -> %.0f frames/s' % (threads, n, dt * 1e3, n / dt))
    fp.close()
