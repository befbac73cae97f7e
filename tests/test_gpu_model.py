"""GPU parity of the product model against the reference's golden vectors — run on the MI355X box with -m gpu."""
import pytest
import torch

import golden_cases as gc
import model_cases as mc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def hip_lib():
    from rat_amd._lib import get_lib
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return get_lib()


@pytest.mark.parametrize("name", [c["name"] for c in gc.CASES])
def test_init_matches_reference_bit_for_bit(name):
    mc.check_init(name, gpu=0)


@pytest.mark.parametrize("name", [c["name"] for c in gc.CASES])
def test_eval_forward(name):
    mc.check_eval(name, gpu=0)


@pytest.mark.parametrize("name", [c["name"] for c in gc.CASES])
def test_two_training_steps(name):
    mc.check_training(name, gpu=0)
