"""Physical constants used on the hot path.

Values restate aimnet/constants.py:6-9 of the reference (CODATA-derived, eV / Angstrom).
"""

Hartree = 27.211386024367243  # eV
half_Hartree = 0.5 * Hartree
Bohr = 0.5291772105638411  # Angstrom
Bohr_inv = 1.0 / Bohr

# k = 1/2 * Hartree * Bohr: the pair-sum prefactor of every Coulomb term that runs over
# ORDERED pairs (lr.py:296; the 1/2 undoes the double count).
COULOMB_FACTOR = half_Hartree * Bohr
