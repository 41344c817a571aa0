"""``ConvLSTMCell`` with the reference's signature and state-dict keys
(/root/reference/model/clstm.py:19-101: Conv1 / Conv2 / Gates .weight/.bias), compute
in libcp360.so (K5): cat(x, h) laid out as one NHWC buffer, three CubePad(1)+3x3
convolutions as MFMA implicit GEMMs with the pad fused into the tile loader, split-K
for the small-M / huge-K shapes, and one gate kernel for
sigmoid/sigmoid/sigmoid/tanh + the cell / hidden update (clstm.py:68-80).
"""
import math

import torch
import torch.nn as nn

from .. import ops, stage_ctx
from .._lib import PRECISIONS as _DTYPES
from .cube_pad import CubePad

KERNEL_SIZE = 3
PADDING = 0



def _stamp(module, extra=()):
    ts = list(module.parameters()) + list(module.buffers())
    return tuple((t.data_ptr(), t._version) for t in ts) + tuple(extra)


class ConvLSTMCell(nn.Module):
    """Generate a convolutional LSTM cell (inference; GPU only)."""

    def __init__(self, input_size, hidden_size, cp=True, precision='fp32'):
        super(ConvLSTMCell, self).__init__()
        if not cp:
            raise NotImplementedError("only cube padding is supported (clstm.py:39-40 would raise NameError)")
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.Conv1 = nn.Conv2d(input_size + hidden_size, 4 * hidden_size, KERNEL_SIZE, padding=PADDING)
        self.Conv2 = nn.Conv2d(4 * hidden_size, 4 * hidden_size, KERNEL_SIZE, padding=PADDING)
        self.Relu = nn.ReLU(inplace=True)
        self.Gates = nn.Conv2d(4 * hidden_size, 4 * hidden_size, KERNEL_SIZE, padding=PADDING)
        self.LSoftMax = nn.LogSoftmax(dim=1)
        self._initialize_weights()
        self.pad = CubePad(1)
        self.precision = precision
        self._plan = None
        self._plan_stamp = None

    def set_precision(self, precision):
        if precision not in _DTYPES:
            raise ValueError("precision must be 'fp32', 'bf16' or 'fp16'")
        self.precision = precision
        return self

    def _initialize_weights(self):          # clstm.py:84-90
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
                if m.bias is not None:
                    m.bias.data.zero_()

    def plans(self):
        dt = _DTYPES[self.precision]
        stamp = _stamp(self, (self.precision,))
        if self._plan is None or stamp != self._plan_stamp:
            dev = self.Conv1.weight.device
            self._plan = {
                'c1': ops.Conv(self.Conv1.weight, None, self.Conv1.bias, 1, 1, True, dt, dev),
                'c2': ops.Conv(self.Conv2.weight, None, self.Conv2.bias, 1, 1, True, dt, dev),
                'g': ops.Conv(self.Gates.weight, None, None, 1, 1, False, dt, dev),   # bias added in the gate kernel
                'gbias': self.Gates.bias.detach().to(device=dev, dtype=torch.float32).contiguous(),
            }
            for k, name in (('c1', 'clstm.Conv1'), ('c2', 'clstm.Conv2'), ('g', 'clstm.Gates')):
                self._plan[k].tag = name
                self._plan[k].clip_only = True      # csrc/ctx.hip packs only the layout the face size's kernel reads
            if dt != torch.float32:
                # the Winograd F(2x2, 3x3) form of the same three convolutions (csrc/wino.hip), packed on first use: taken when
                # the launch has enough tiles (cp360_wino_preferred: 4 clips of 7x7 faces, one clip of 16x16 faces, ...)
                self._plan['w1'] = ops.WinoConv(self.Conv1.weight, self.Conv1.bias, True, dt, dev)
                self._plan['w2'] = ops.WinoConv(self.Conv2.weight, self.Conv2.bias, True, dt, dev)
                self._plan['wg'] = ops.WinoConv(self.Gates.weight, None, False, dt, dev)
                for k, name in (('w1', 'clstm.Conv1'), ('w2', 'clstm.Conv2'), ('wg', 'clstm.Gates')):
                    self._plan[k].tag = name
            self._plan_stamp = stamp
        return self._plan

    def uses_winograd(self, n6, w):
        """True when a cell update on n6 faces of w x w runs in the Winograd domain (the library's rule, on Conv2's shape)."""
        p = self.plans()
        return 'w2' in p and p['w2'].preferred(n6, w)

    def step_nhwc(self, xh, c_prev, c_next, h_f32=None, bufs=None, x_next=None):
        """One cell update on the fused layout.
        xh     [6B, w, w, Cin+Ch]  compute dtype: channels [0, Cin) = input frame,
               [Cin, Cin+Ch) = previous hidden; the NEW hidden is written back into
               the hidden half (ready for the next step)
        c_prev / c_next [6B, w, w, Ch] f32; h_f32 optional f32 copy of the new hidden.
        x_next (ClipRunner): the next frame's window normalisation rides on the gate kernel (needs Cin == Ch).
        """
        if stage_ctx.USE_CTX:
            # ONE C call per cell update (cp360_clstm_step, csrc/ctx.hip); CP360_CTX=0 plans the same launches here
            stage = self.__dict__.get('_stage')
            if stage is None:
                stage = self.__dict__['_stage'] = stage_ctx.ClstmStage(self)
            stage.step(xh, c_prev, c_next, h_f32, x_next)
            return
        p = self.plans()
        n6, w, _, _ = xh.shape
        # x_next = (cam, minmax, P, clip_stride, t_next): the gate kernel also writes the next step's normalised input
        xn = None if x_next is None else (x_next[0], x_next[1], 0, x_next[2], x_next[3], x_next[4])
        if self.uses_winograd(n6, w):
            # (the same launches as csrc/ctx.hip's clstm_run: between two convolutions ONE fused output + input transform where
            # the faces allow it, else the two kernels through an activation buffer)
            convs = (p['w1'], p['w2'], p['wg'])
            v, d = convs[0].input(xh)
            for k in range(2):
                m = convs[k].gemm(v, d)
                fused = convs[k].output_input(m, d, convs[k + 1])
                if fused is None:
                    a = convs[k].output(m, d, out=None if bufs is None else bufs[k])
                    fused = convs[k + 1].input(a)
                v, d = fused
            m = convs[2].gemm(v, d)
            convs[2].gates_from(m, d, p['gbias'], c_prev, c_next, xh, self.input_size, h_f32, x_next=xn)
            return
        a1 = p['c1'](xh, out=None if bufs is None else bufs[0])
        a2 = p['c2'](a1, out=None if bufs is None else bufs[1])
        sr = (4 * self.hidden_size) % 32 == 0              # gate slabs in packed-row order (64-byte stores)
        partial, splits = p['g'](a2, raw_f32=True, slab_rows=sr)
        ops.lstm_gates(partial, splits, p['gbias'], c_prev, c_next, xh, self.input_size, h_f32,
                       n6 * w * w, self.hidden_size, slab_rows=sr, x_next=xn)

    def forward(self, input_, prev_state=None):
        """input_ [6B, Cin, w, w]; prev_state = (hidden, cell) [6B, Ch, w, w] or None
        (zeros, clstm.py:47-52).  Returns (hidden, cell) as f32 NCHW tensors."""
        with torch.no_grad():
            ops.require_gpu(input_)
            dt = _DTYPES[self.precision]
            n6, cin, w, w2 = input_.shape
            ch = self.hidden_size
            dev = input_.device
            xh = torch.empty((n6, w, w2, cin + ch), dtype=dt, device=dev)
            ops.nchw_to_nhwc(input_.float(), out=xh, coff=0)
            if prev_state is None:
                xh[..., cin:].zero_()
                c_prev = torch.zeros((n6, w, w2, ch), dtype=torch.float32, device=dev)
            else:
                prev_hidden, prev_cell = prev_state
                ops.nchw_to_nhwc(prev_hidden.to(dev).float(), out=xh, coff=cin)
                c_prev = ops.nchw_to_nhwc(prev_cell.to(dev).float())
            c_next = torch.empty_like(c_prev)
            h_f32 = torch.empty_like(c_prev)
            self.step_nhwc(xh, c_prev, c_next, h_f32)
            return ops.nhwc_to_nchw(h_f32), ops.nhwc_to_nchw(c_next)

    def load_pretrained_model_seq(self, pretrained_state_dict):
        """Copy parameters BY POSITION (clstm.py:92-101)."""
        custom_state_dict = self.state_dict()
        for name, param in zip(custom_state_dict.keys(), pretrained_state_dict.values()):
            if isinstance(param, nn.Parameter):
                param = param.data
            try:
                custom_state_dict[name].copy_(param)
            except Exception:
                print("skip loading key '{}' due to inconsistent size".format(name))
        self.load_state_dict(custom_state_dict)
