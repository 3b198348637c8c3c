import sys; sys.path.insert(0, ".")
import torch
from openjph_amd import codec
from openjph_amd.plan import make_params

frame = torch.randint(0, 4096, (3, 4320, 7680), dtype=torch.int16, device="cuda")   # 12-bit samples
enc = codec.Encoder(make_params(7680, 4320, 3, bit_depth=12, reversible=False, qstep=0.001))
enc.run_device(frame)
codestream = enc.finish()
dec = codec.Decoder(codestream)
out = dec.run_device(dtype=torch.int16)
torch.cuda.synchronize()
print(len(codestream), out.shape, out.dtype, int((out.int() - frame.int()).abs().max()))
