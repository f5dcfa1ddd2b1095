"""The DEFAULT route of the product under the tight gradient check (GPU).

tests/test_step_gpu.py holds the gradients to 1e-4 against the fp64 oracle evaluated at the HIP path's own ReLU / max-pool decisions —
but it reads those decisions through srlz.hotpath.TAPS, and TAPS switches the route: no PoolLink (BatchNorm-backward sums out of the
next convolution's data-gradient epilogue), no deferred decoder BatchNorm backward, hence no fused ConvTranspose block backward, and
no loss inside the last ConvTranspose.  Here the step is `SRL4robotics.trainStep` exactly as `learn()` / `bench.py` run it (batched
pair with two BatchNorm groups, every fusion on), observed through srlz.hotpath.OBSERVE, which only keeps references to tensors the
default route saves anyway: the pooled maps with their argmax bytes, the decoder blocks' raw inputs with their BatchNorm records.
The comparison is on the gradient BUCKET Adam consumes (srlz.optim.FlatParams.grad), parameter by parameter, at 1e-4 relative.

Reference: the loop body models/learner.py:373-497 of /root/reference (restated by oracle/torch_twin.py::train_step).
"""
import pytest

from route_check import check_default_route_bucket

pytestmark = pytest.mark.gpu
RTOL = 1e-4

CASES = [("step_ae_b4", ["autoencoder"], 4, 3),
         ("step_vae_b4", ["vae"], 4, 3),
         ("step_aeif_b2", ["autoencoder", "inverse", "forward"], 2, 3),
         ("step_ae_c6_b2", ["autoencoder"], 2, 6)]


@pytest.mark.parametrize("name,losses,B,C", CASES)
def test_default_route_gradient_bucket_matches_decision_pinned_fp64_oracle(name, losses, B, C):
    check_default_route_bucket(losses, B, C, rtol=RTOL)
