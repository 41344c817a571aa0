"""CPU tests of the module surface that needs no kernel: ``resnet50(pretrained=True)`` follows
/root/reference/model/resnet_cubic.py:228-237 (``model_zoo.load_url`` + ``load_pretrained_model``), ``im_norm`` /
``sigmoid`` follow /root/reference/utils/utils.py:28-37."""
import os

import numpy as np
import pytest
import torch

from cp_360_weakly_supervised_saliency_amd.model import resnet_cubic
from cp_360_weakly_supervised_saliency_amd.utils import synth
from cp_360_weakly_supervised_saliency_amd.utils.utils import im_norm, sigmoid


def test_pretrained_true_without_network_raises_naming_the_url(tmp_path, monkeypatch):
    """No network and an empty hub cache: a RuntimeError that names the URL - never silently random weights."""
    monkeypatch.setenv('TORCH_HOME', str(tmp_path))
    with pytest.raises(RuntimeError) as e:
        resnet_cubic.resnet50(pretrained=True)
    assert 'download.pytorch.org/models/resnet50-19c8e357.pth' in str(e.value)


def test_pretrained_true_loads_a_cached_checkpoint(tmp_path, monkeypatch):
    """A checkpoint in torch's hub cache is loaded by name exactly as the reference does (load_pretrained_model:
    copy by key, skip nothing here): every parameter and BatchNorm buffer of the model equals the file's."""
    monkeypatch.setenv('TORCH_HOME', str(tmp_path))
    ck = os.path.join(str(tmp_path), 'hub', 'checkpoints')
    os.makedirs(ck)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.resnet50_state(seed=1).items()}
    torch.save(sd, os.path.join(ck, os.path.basename(resnet_cubic.model_urls['resnet50'])))
    m = resnet_cubic.resnet50(pretrained=True)
    own = m.state_dict()
    for k, v in sd.items():
        assert torch.equal(own[k], v), k
    # an unknown key raises KeyError like the reference (resnet_cubic.py:189-191)
    with pytest.raises(KeyError):
        m.load_pretrained_model({'not.a.key': torch.zeros(1)})


def test_im_norm_in_place_and_sigmoid():
    r = np.random.RandomState(0)
    img = r.rand(5, 7, 3)
    want = (img - np.array([0.485, 0.456, 0.406])) / np.array([0.229, 0.224, 0.225])
    out = im_norm(img, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    assert out is img and np.array_equal(out, want)                  # in place, same float64 operations
    assert np.allclose(sigmoid(np.array([0.0, 1.0])), [0.5, 1 / (1 + np.exp(-1.0))])


def test_stage_stamp_sees_replaced_parameter_objects():
    """stage_ctx._stamp (the key that decides when a stage context re-packs its weights) caches the module's tensor list; it
    must still change when a Parameter / submodule OBJECT is replaced after the first forward (``model.fc = nn.Linear(..)``,
    ``m.weight = Parameter(..)``), when weights are updated in place, and when ``_apply`` overwrites the parameters
    (torch.__future__.set_overwrite_module_params_on_conversion) - and stay equal otherwise."""
    import torch
    import torch.nn as nn
    from cp_360_weakly_supervised_saliency_amd import stage_ctx as sc
    net = nn.Sequential(nn.Conv2d(3, 4, 1), nn.BatchNorm2d(4), nn.Linear(4, 2))
    a = sc._stamp(net)
    assert sc._stamp(net) == a and sc._stamp(net, ('x',)) != a
    net[2] = nn.Linear(4, 2)
    b = sc._stamp(net)
    assert b != a
    net[0].weight = nn.Parameter(torch.zeros(4, 3, 1, 1))
    c = sc._stamp(net)
    assert c != b
    with torch.no_grad():
        net[0].weight.add_(1)
    d = sc._stamp(net)
    assert d != c
    net.load_state_dict({k: v.clone() for k, v in net.state_dict().items()})
    e = sc._stamp(net)
    assert e != d
    torch.__future__.set_overwrite_module_params_on_conversion(True)
    try:
        net.double()
    finally:
        torch.__future__.set_overwrite_module_params_on_conversion(False)
    f = sc._stamp(net)
    assert f != e and sc._stamp(net) == f
