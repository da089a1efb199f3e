"""TEST INFRASTRUCTURE ONLY: CPU oracle of the SHT hot path (see oracle/sht_oracle.py, oracle/sht_oracle.c).
Nothing under plancklens_amd/ imports this package."""
