"""Hot-path operators.  Every op runs its hand-written gfx950 kernel when its inputs live on
the GPU and a PyTorch fp32 reference when they live on the host; the reference is also the
numerics oracle of the kernel tests."""
from . import fitness
