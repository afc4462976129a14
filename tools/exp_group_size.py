import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vec_vad_amd.bank import UNetBank, UnitSpec
B = 256
g = torch.Generator().manual_seed(0)
raw = torch.randint(0, 256, (1024, 5, 32, 32, 3), dtype=torch.uint8, generator=g).cuda()
flow = (torch.randn(1024, 1, 32, 32, 2, generator=g) * 2).cuda()
for units in ([UnitSpec('raw', 4, 4)], [UnitSpec('raw', i, i) for i in range(2)], [UnitSpec('raw', i, i) for i in range(3)],
              [UnitSpec('raw', i, i) for i in range(5)] + [UnitSpec('of', 4, 0)]):
    bank = UNetBank(units, nf=32, device='cuda')
    with torch.no_grad():
        bank.params.normal_(0, 0.05)
        for k, (off, shp) in bank.lay.p.items():
            if k.endswith('.g'):
                n = 1
                for s_ in shp: n *= s_
                bank.params[:, off:off + n] = 1.0
    idx = torch.arange(B).cuda()
    def step():
        ws = bank.set_input_cubes(raw, flow, idx, B)
        bank.forward(ws, True)
        bank.backward(ws)
        bank.adam_step()
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print('G=%d: %.3f ms/step -> %.3f ms per UNet' % (len(units), dt * 1e3, dt * 1e3 / len(units)), flush=True)
