import torch, sys
sys.path.insert(0, '.')
from tests.util import golden_inputs, maxabs
from bde2vid_amd import canonical
from bde2vid_amd.model import build_model
from bde2vid_amd.weights import formula_state_dict
cfg = canonical()
m = build_model(cfg, formula_state_dict(cfg), 'cuda:0')
xs = golden_inputs(6, 1, 5, 184, 240, 2468)
inp = [{'events': torch.from_numpy(x).cuda()} for x in xs]
with torch.no_grad():
    base = torch.stack(m(inp)).clone()
    b2 = torch.stack(m(inp)).clone()
    print('repeat', maxabs(b2, base))
    for key, off, on in (('lstm_sbk', 0, 1), ('winblock_sb', 0, 1), ('wide_fuse_qkv', 0, 1), ('lstm_two_streams', 1, 0), ('lstm_fuse_x', 0, 1), ('sb_terms', 3, 2), ('conv_sb', 0, 1)):
        m.set_tuning(key, off)
        y = torch.stack(m(inp)).clone()
        yy = torch.stack(m(inp)).clone()
        m.set_tuning(key, on)
        d = torch.stack(m(inp)).clone()
        d2 = torch.stack(m(inp)).clone()
        diff = (d - base).abs()
        print(key, 'off-vs-base', maxabs(y, base), 'default-after', float(diff.max()), 'again', maxabs(d2, base), 'frames', [float(diff[t].max()) for t in range(6)])
