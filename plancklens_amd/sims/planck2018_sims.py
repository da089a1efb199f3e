"""FFP10 input libraries of plancklens/sims/planck2018_sims.py:160-250 (`cmb_len_ffp10`, `cmb_unl_ffp10`): same hash
dictionaries and file layout under $CFS (the NERSC project directory).  Constructing them costs nothing -- that is all a
parameter file needs at import (params/idealized_example.py:64) -- and the getters read the public FFP10 alm files when
$CFS points to them."""
import os

from .. import hp

_LEN = 'cmb/data/generic/cmb/ffp10/mc/scalar/ffp10_lensed_scl_cmb_000_alm_mc_%04d.fits'
_UNL = 'cmb/data/generic/cmb/ffp10/mc/scalar/ffp10_unlensed_scl_cmb_000_tebplm_mc_%04d.fits'


def _read(rel, idx, hdu):
    if 'CFS' not in os.environ:
        raise RuntimeError('the FFP10 simulations live in the NERSC project directory: set $CFS to its root')
    return hp.read_alm(os.path.join(os.environ['CFS'], rel % idx), hdu=hdu)


class cmb_len_ffp10(object):
    """lensed scalar CMB alms (muK) of the FFP10 Monte Carlo set"""

    def __init__(self):
        pass

    def hashdict(self):
        return {'sim_lib': 'ffp10 lensed scalar cmb inputs, freq 0'}

    @staticmethod
    def get_sim_tlm(idx):
        return 1e6 * _read(_LEN, idx, 1)

    @staticmethod
    def get_sim_elm(idx):
        return 1e6 * _read(_LEN, idx, 2)

    @staticmethod
    def get_sim_blm(idx):
        return 1e6 * _read(_LEN, idx, 3)


class cmb_unl_ffp10(object):
    """unlensed scalar CMB alms (muK) and the lensing potential of the FFP10 Monte Carlo set"""

    def __init__(self):
        pass

    def hashdict(self):
        return {'sim_lib': 'ffp10 unlensed scalar cmb inputs'}

    @staticmethod
    def get_sim_tlm(idx):
        return 1e6 * _read(_UNL, idx, 1)

    @staticmethod
    def get_sim_elm(idx):
        return 1e6 * _read(_UNL, idx, 2)

    @staticmethod
    def get_sim_blm(idx):
        return 1e6 * _read(_UNL, idx, 3)

    @staticmethod
    def get_sim_plm(idx):
        return _read(_UNL, idx, 4)
