"""Debug aid: where do the packed and the padded layout differ?  usage: python tools/packed_dbg.py [conformer 0/1] [layers]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_packed_rows_gpu as T
from s2t_amd import s2t_transformer as M

conf = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 1
model = T._model(conf, enc_layers=nl, dec_layers=1)
model.eval()
sample, lens = T._sample(24, 1000, 3)
ni = sample["net_input"]
outs = {}
with torch.no_grad():
    for packed in (False, True):
        with T._layout(packed):
            enc = model.encoder(src_tokens=ni["src_tokens"], src_lengths=ni["src_lengths"])
            outs[packed] = enc["encoder_out"][0].float()
a, b = outs[False], outs[True]
sub = model.encoder.subsample.get_out_seq_lens_tensor(torch.tensor(lens)).tolist()
d = (a - b).abs().amax(-1)  # T x B
print("T' =", a.shape[0], "layers", nl, "conformer", conf)
for bi, l in enumerate(sub):
    col = d[:l, bi]
    bad = (col > 0).nonzero().flatten().tolist()
    print("utt %2d len %3d  max diff %.4g  #bad %d  first %s last %s" % (bi, l, float(col.max()), len(bad), bad[:6], bad[-6:]))
