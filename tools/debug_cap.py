import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from _checkers import Checker, Params
import steganosaurus_amd as S
from steganosaurus_amd.synth import cover_rgb
orc = Checker('orc')
img = cover_rgb(64, 64, 0)
ctx = S.Context(64, 64)
for it in range(3):
    ctx.forward_rgb8(img, 0)
    m = ctx.medians()
    print('med', m, orc.forward_rgb8(img, 0)[1])
    print('cap thr=0.01med', ctx.capacity(0.01 * m), 'thr=0', ctx.capacity(np.zeros(3)), 'want', orc.capacity_rgb8(img)[0])
    for p in range(3):
        t = np.full(3, 1e30); t[p] = 0.0
        print('  only plane', p, ctx.capacity(t))
