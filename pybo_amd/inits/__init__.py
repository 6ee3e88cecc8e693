"""Initial designs; the exported names match pybo.inits (`init_middle|uniform|latin|sobol`), plus the
device-resident grids `init_sobol_device` / `init_uniform_device` (-> `DeviceGrid`)."""
from .methods import (init_middle, init_uniform, init_latin, init_sobol, init_sobol_device,
                      init_uniform_device)
from .._lib import DeviceGrid, ShardedDeviceGrid

__all__ = ['init_middle', 'init_uniform', 'init_latin', 'init_sobol', 'init_sobol_device',
           'init_uniform_device', 'DeviceGrid', 'ShardedDeviceGrid']
