"""plancklens hot path on AMD MI355X (gfx950): spin-0..3 HEALPix SHTs, quadratic estimators and the
qcinv conjugate-gradient Wiener filter behind the reference's plancklens.shts / filt / qest / qecl API."""
__version__ = '0.1.0'
