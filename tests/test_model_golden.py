"""End-to-end parity of the PRODUCT model (rat_amd.RAT_m2: host code + C-ABI + kernel sources) against the golden
vectors the real reference produced.  On CPU the kernels run through the host-emulation build (tests/emu); the same
checks run on the MI355X with the HIP build in tests/test_gpu_model.py."""
import os
import sys

import numpy as np
import pytest
import torch

import golden_cases as gc
import model_cases as mc

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


@pytest.fixture(scope="module", autouse=True)
def emu_lib():
    import build_emu
    import rat_amd._lib as L
    old = L._default
    L._default = L.RatLib(build_emu.build())
    yield L._default
    L._default = old


@pytest.mark.parametrize("name", ["tiny_seq_bn", "bare_no_proj", "mltag_shape"])
def test_init_matches_reference_bit_for_bit(name):
    mc.check_init(name, gpu=-1)


# the emulator runs one OS thread per GPU thread: keep the CPU suite to the small cases (the GPU suite runs them all)
@pytest.mark.parametrize("name", ["tiny_seq_bn", "bare_no_proj", "mltag_shape", "tmall_shape"])
def test_eval_forward(name):
    mc.check_eval(name, gpu=-1)


@pytest.mark.parametrize("name", ["tiny_seq_bn", "bare_no_proj"])
def test_two_training_steps(name):
    mc.check_training(name, gpu=-1)
