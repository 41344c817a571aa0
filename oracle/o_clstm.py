"""Oracle: ConvLSTM cell and the sliding-window clip semantics (torch-CPU fp32).

Cell follows /root/reference/model/clstm.py:42-82; the window follows
/root/reference/temporal_model/test_temporal.py:57-85.
"""
import numpy as np
import torch
import torch.nn.functional as Fn

from .o_resnet import cubepad_t
from .o_c2e import saliency_from_hidden


def clstm_step(x, hidden, cell, sd):
    """clstm.py:42-82.  x, hidden, cell: torch float32 [6B, Ch, w, w].

    cat(x, h) -> [CubePad(1) -> conv3x3 (+bias)] x3 with ReLU after the first two
    (:56-64); gates chunked in the order in, remember, out, cell (:68); sigmoid on
    the first three, tanh on the fourth (:71-76);
    cell' = remember*cell + in*cell_gate, hidden' = out*tanh(cell') (:79-80).
    """
    with torch.no_grad():
        out = torch.cat((x, hidden), 1)
        out = Fn.relu(Fn.conv2d(cubepad_t(out, 1), sd['Conv1.weight'], sd['Conv1.bias']))
        out = Fn.relu(Fn.conv2d(cubepad_t(out, 1), sd['Conv2.weight'], sd['Conv2.bias']))
        gates = Fn.conv2d(cubepad_t(out, 1), sd['Gates.weight'], sd['Gates.bias'])
        i_g, f_g, o_g, c_g = gates.chunk(4, 1)
        i_g, f_g, o_g = torch.sigmoid(i_g), torch.sigmoid(f_g), torch.sigmoid(o_g)
        c_g = torch.tanh(c_g)
        cell = f_g * cell + i_g * c_g
        hidden = o_g * torch.tanh(cell)
    return hidden, cell


def window_hidden(frames, sd, all_steps=False):
    """test_temporal.py:63-80 for ONE window.

    frames: ndarray [T, 6, C, w, w] float32 (the T cube_feat arrays of the window).
    mn/mx over the whole window (:66-67); hidden = cell = normalised frame 0
    (:70-73); all T frames (frame 0 again first) are fed in order (:76-79);
    returns the final hidden [6, C, w, w] float32 (all_steps: the hidden after every step, [T, 6, C, w, w]).
    """
    frames = np.asarray(frames, dtype=np.float32)
    mx, mn = np.max(frames), np.min(frames)
    init = (frames[0] - mn) / (mx - mn)
    hidden = torch.from_numpy(init.astype(np.float32))
    cell = torch.from_numpy(init.astype(np.float32))
    trace = []
    for t in range(frames.shape[0]):
        f = torch.from_numpy(((frames[t] - mn) / (mx - mn)).astype(np.float32))
        hidden, cell = clstm_step(f, hidden, cell, sd)
        if all_steps:
            trace.append(hidden.numpy().copy())
    return np.stack(trace) if all_steps else hidden.numpy()


def window_saliency(frames, sd, face_map=None, out_coord=None, align_corners=False):
    """test_temporal.py:57-85 for one window: -> saliency [2w, 4w] float32."""
    return saliency_from_hidden(window_hidden(frames, sd), face_map, out_coord, align_corners)
