"""`python -m stodynprog_amd`: the energy-storage demo the reference runs as
`python stodynprog.py` (reference stodynprog/stodynprog.py:879-965): a NaS
storage following a request modelled as an AR(1) process, two value
iterations -- here on the GPU, without the plot."""
import numpy as np

from . import models


def main():
    sys_, dpsolv = models.nas_demo()
    sys_.print_summary()
    print('')
    dpsolv.print_summary()
    print('')
    print('Running 2 value iterations...')
    J_N = np.zeros(dpsolv._state_grid_shape)
    J, u = dpsolv.value_iteration(J_N)
    J, u = dpsolv.value_iteration(J)
    print('optimal stored power after 2 iterations: min {:.3f}, max {:.3f} MW'.format(
        u[..., 0].min(), u[..., 0].max()))
    info = dpsolv.backend_info
    print('kernel: {}, {} controls at most'.format(info.get('kernel'), info.get('max_controls')))
    return J, u


if __name__ == '__main__':
    main()
