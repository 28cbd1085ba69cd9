"""The single-launch preparation of small periodic batches (csrc/nlist.hip, prep_small_kernel; engine option "prep_fused") against
the seven separate kernels it replaces: the neighbour rows, hence every output, must be bitwise identical, and the input sanity
flags must be raised the same way."""
from __future__ import annotations

import numpy as np
import pytest
import torch

import test_gpu_parity as P
from conftest import golden

pytestmark = pytest.mark.gpu


def both(eng, fn, option="prep_fused"):
    out = []
    try:
        for v in (1, 0):
            eng.set_option(option, v)
            out.append(fn())
    finally:
        eng.set_option(option, 1)
    return out


@pytest.mark.parametrize("name", ["pbc96_dsf15", "pbc96_dsf8_wrapped", "pbc2x96_dsf9"])
def test_fixtures_bitwise(hip_engine, name):
    g = golden(name)
    kw = dict(dsf_rc=float(g["dsf_rc"]), dsf_alpha=float(g["dsf_alpha"]))
    a, b = both(hip_engine, lambda: P.run(hip_engine, g, "dsf", stress=True, **kw)[0])
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    P.compare(a, g, 96, name + "/reference golden (fused preparation)")


@pytest.mark.parametrize("dftd3", [False, True])
def test_supercell_bitwise(hip_engine, dftd3):
    """2 016 atoms in a triclinic-free but many-bin cell, atoms displaced across the cell faces (wrapping), with and without the
    species pass the D3 term needs."""
    from aimnetcentral_amd import workloads

    c, z, cell = workloads.glucose_supercell((3, 2, 3) if not dftd3 else (2, 2, 2))
    rng = np.random.default_rng(11)
    c = (c + rng.normal(0.0, 0.03, c.shape) + rng.integers(-1, 2, (len(z), 3)) @ cell).astype(np.float32)
    dev = hip_engine.device
    kw = {}
    if dftd3:
        par, tables = P._d3(9.0)
        hip_engine.set_dftd3_tables(tables)
        kw["dftd3"] = par

    def go():
        r = hip_engine.eval(torch.from_numpy(c).to(dev), torch.from_numpy(z).to(dev), torch.zeros(len(z), dtype=torch.int64, device=dev),
                            torch.zeros(1, device=dev), cell=torch.from_numpy(cell.astype(np.float32)).to(dev), forces=True, stress=True,
                            coulomb="dsf", dsf_rc=9.0, **kw)
        return {k: v.cpu().numpy() for k, v in r.items()}

    a, b = both(hip_engine, go)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert np.isfinite(a["energy"]).all() and np.abs(a["forces"]).max() > 0


@pytest.mark.parametrize("seed", [1, 5, 9, 13, 17, 21, 25, 29])
def test_strained_triclinic_cells_bitwise(hip_engine, seed):
    """The periodic cases of the randomised sweep (strained cells, vacancies, atoms moved out of the box by lattice vectors, one or
    two systems, mixed periodicity): wrapping and binning use one pinned arithmetic form (cellwalk.h, cell_frac / wrap_into_cell),
    so the two preparations agree to the bit here too - with the compiler's own contraction choices they differed in the last bit
    of the wrapped coordinates, which these hot geometries turn into 1e-5 eV."""
    import test_gpu_fuzz as Z

    case = Z.make_case(seed)
    assert "cell" in case[6]
    a, b = both(hip_engine, lambda: Z.run_case(hip_engine, case))
    for k in a:
        assert np.array_equal(a[k], b[k]), (k, case[-1])


@pytest.mark.parametrize("name,coulomb", [("taxol", "simple"), ("batch5", "simple"), ("batch5", "dsf")])
def test_molecules_bitwise(hip_engine, name, coulomb):
    """Molecules (no cell): the single launch replaces the status memset, the molecule-offset pass and the coordinate copy."""
    g = golden(name)
    a, b = both(hip_engine, lambda: P.run(hip_engine, g, coulomb)[0])
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_many_molecules_and_a_large_one_bitwise(hip_engine):
    """300 molecules in one batch (more systems than the periodic form of the kernel takes) and one 2 000-atom molecule (bounding-box
    cell list behind the single launch)."""
    from aimnetcentral_amd import workloads

    dev = hip_engine.device
    c, z, mol, q = workloads.random_batch(300, 3, 12, seed=5)
    big_c, big_z = workloads.random_organic(2000, np.random.default_rng(3))
    for cc, zz, mm, qq in ((c, z, mol, q), (big_c.astype(np.float32), big_z, np.zeros(2000, dtype=np.int64), np.zeros(1, np.float32))):
        assert len(zz) <= 4096

        def go():
            r = hip_engine.eval(torch.from_numpy(cc).to(dev), torch.from_numpy(zz).to(dev), torch.from_numpy(mm).to(dev), torch.from_numpy(qq).to(dev),
                                forces=True, coulomb="simple")
            return {k: v.cpu().numpy() for k, v in r.items()}

        a, b = both(hip_engine, go)
        for k in a:
            assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("reps", [(1, 1, 1), (2, 3, 4)])
def test_energy_sums_riding_on_the_stress_launches_bitwise(hip_engine, reps):
    """Engine option "energy_rides": with a stress request the molecule energy sums run as riders of the two stress launches instead
    of two launches of their own - the same partial sums in the same order (96 atoms: one slice; 2 304 atoms: five slices)."""
    from aimnetcentral_amd import workloads

    c, z, cell = workloads.glucose_supercell(reps)
    dev = hip_engine.device

    def go():
        r = hip_engine.eval(torch.from_numpy(c.astype(np.float32)).to(dev), torch.from_numpy(z).to(dev),
                            torch.zeros(len(z), dtype=torch.int64, device=dev), torch.zeros(1, device=dev),
                            cell=torch.from_numpy(cell.astype(np.float32)).to(dev), forces=True, stress=True, coulomb="dsf", dsf_rc=9.0)
        return {k: v.cpu().numpy() for k, v in r.items()}

    # (the whole-molecule form of the riding sums, "sums_whole", adds the same fp64 terms in another association - equal to 1e-13
    # relative, checked below; the bitwise claim is about the sliced sums)
    hip_engine.set_option("sums_whole", 0)
    try:
        a, b = both(hip_engine, go, "energy_rides")
    finally:
        hip_engine.set_option("sums_whole", 1)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert np.isfinite(a["energy"]).all() and a["energy"][0] < 0
    w = go()  # defaults: energy sums riding, whole-molecule blocks
    assert np.abs(w["energy"] - a["energy"]).max() <= 1e-13 * np.abs(a["energy"]).max()
    for k in a:
        if k == "stress":  # (its sums change association too: the last float bit)
            assert np.allclose(w[k], a[k], rtol=1e-6, atol=1e-12), k
        elif k != "energy":
            assert np.array_equal(w[k], a[k]), k


@pytest.mark.parametrize("option", ["setup_rides", "status_rides"])
@pytest.mark.parametrize("seed", [1, 5, 9])
def test_rider_forms_bitwise(hip_engine, option, seed):
    """The cell + bin-grid setup as a rider block of the molecule-offset launch (atom counts by binary search in mol_idx), and the
    list's status words as rider blocks of the SR-Coulomb launch: identical results on strained cells, one or two systems of
    different sizes; the separate preparation kernels are the ones that run (prep_fused = 0)."""
    import test_gpu_fuzz as Z

    case = Z.make_case(seed)
    hip_engine.set_option("prep_fused", 0)
    try:
        a, b = both(hip_engine, lambda: Z.run_case(hip_engine, case), option)
    finally:
        hip_engine.set_option("prep_fused", 1)
    for k in a:
        assert np.array_equal(a[k], b[k]), (k, case[-1])
    assert np.array_equal(hip_engine.last_status[:7], hip_engine.last_status[:7])


def test_status_words_identical_with_and_without_riders(hip_engine):
    """Longest row and overflow flag arrive the same whichever launch reduces them; an overflowing capacity still grows."""
    g = golden("pbc96_dsf15")
    st = []
    for v in (1, 0):
        hip_engine.set_option("status_rides", v)
        try:
            hip_engine.max_nb = 16  # too small: the first evaluation overflows, the engine grows the rows and repeats
            P.run(hip_engine, g, "dsf", dsf_rc=9.0)
            st.append((hip_engine.last_status.copy(), hip_engine.max_nb))
        finally:
            hip_engine.set_option("status_rides", 1)
    assert np.array_equal(st[0][0], st[1][0]) and st[0][1] == st[1][1] and st[0][0][2] == 0 and st[0][0][0] > 16


@pytest.mark.parametrize("name,coulomb", [("taxol", "simple"), ("batch5", "simple"), ("batch5", "dsf")])
def test_energy_sums_and_charge_copy_riding_on_the_force_launch_bitwise(hip_engine, name, coulomb):
    """Forces-only evaluations of small molecules: the molecule energy sums and the copy of the charges into the output ride on the
    force-negation launch at the end (option "energy_rides")."""
    g = golden(name)
    a, b = both(hip_engine, lambda: P.run(hip_engine, g, coulomb)[0], "energy_rides")
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    if name == "taxol":
        P.compare(a, g, 113, "taxol/reference golden (riders)")


@pytest.mark.parametrize("name", ["pbc96_dsf15", "pbc2x96_dsf9"])
def test_whole_cell_sums_agree_with_the_sliced_ones(hip_engine, name):
    """Option "sums_whole": one block per cell / per molecule sums the virial / the energy and writes the result (no slices, no finish
    launch) when the sums ride beside the force gather.  A different association of fp64 sums: energies to 1e-13 relative, stress
    to the last float bit or two; forces and charges untouched."""
    from aimnetcentral_amd import workloads

    if name == "pbc96_dsf15":  # above split_max, so that the reverse-pair force gather (the launch the sums ride on) is what runs
        c, z, cell = workloads.glucose_supercell((2, 3, 4))
        mol, q = np.zeros(len(z), dtype=np.int64), np.zeros(1, np.float32)
    else:
        g = golden(name)
        c, z, cell, mol, q = g["coord"], g["numbers"], g["cell"], g["mol_idx"], np.atleast_1d(g["charge"]).astype(np.float32)
        c, z, mol = np.tile(c, (8, 1)), np.tile(z, 8), np.repeat(np.arange(16), 96)  # 16 systems of 96 atoms: 1 536 atoms
        cell, q = np.tile(cell, (8, 1, 1)), np.tile(q, 8)
    dev = hip_engine.device

    def go():
        r = hip_engine.eval(torch.from_numpy(c.astype(np.float32)).to(dev), torch.from_numpy(z).to(dev), torch.from_numpy(mol).to(dev),
                            torch.from_numpy(q).to(dev), cell=torch.from_numpy(cell.astype(np.float32)).to(dev), forces=True, stress=True,
                            coulomb="dsf", dsf_rc=9.0)
        return {k: v.cpu().numpy() for k, v in r.items()}

    a, b = both(hip_engine, go, "sums_whole")
    assert np.array_equal(a["forces"], b["forces"]) and np.array_equal(a["charges"], b["charges"])
    assert np.abs(a["energy"] - b["energy"]).max() <= 1e-13 * np.abs(b["energy"]).max()
    assert np.abs(a["stress"] - b["stress"]).max() <= 3e-7 * np.abs(b["stress"]).max()


def test_status_array_without_a_memset(hip_engine):
    """Option "status_owned": nothing zeroes the status words in front of the evaluation, one rider block stores all eight - the same
    words arrive, bad inputs are still flagged (their flags travel per wave), an overflowing row capacity still grows.  4 608 atoms:
    above the single-launch preparation, so the separate kernels and their riders are what runs."""
    from aimnetcentral_amd import workloads

    c, z, cell = workloads.glucose_supercell((4, 3, 4))
    dev = hip_engine.device
    args = [torch.from_numpy(c.astype(np.float32)).to(dev), torch.from_numpy(z).to(dev), torch.zeros(len(z), dtype=torch.int64, device=dev),
            torch.zeros(1, device=dev)]
    kw = dict(cell=torch.from_numpy(cell.astype(np.float32)).to(dev), forces=True, stress=True, coulomb="dsf", dsf_rc=9.0)
    out = []
    for v in (1, 0):
        hip_engine.set_option("status_owned", v)
        try:
            hip_engine.max_nb = 16
            r = hip_engine.eval(*args, **kw)
            out.append(({k: t.cpu().numpy() for k, t in r.items()}, hip_engine.last_status.copy(), hip_engine.max_nb))
            zb = z.copy()
            zb[4000] = 99
            with pytest.raises(ValueError, match="atomic number"):
                hip_engine.eval(args[0], torch.from_numpy(zb).to(dev), args[2], args[3], **kw)
            mb = np.zeros(len(z), dtype=np.int64)
            mb[100] = 1
            with pytest.raises(ValueError):
                hip_engine.eval(args[0], args[1], torch.from_numpy(mb).to(dev), args[3], **kw)
        finally:
            hip_engine.set_option("status_owned", 1)
    for k in out[0][0]:
        assert np.array_equal(out[0][0][k], out[1][0][k]), k
    assert np.array_equal(out[0][1], out[1][1]) and out[0][2] == out[1][2] and out[0][1][0] > 16 and not out[0][1][1:].any()


def test_bad_inputs_are_flagged_the_same(hip_engine):
    g = golden("pbc96_dsf15")
    dev = hip_engine.device
    cell = torch.from_numpy(g["cell"]).to(dev)
    coord = torch.from_numpy(g["coord"]).to(dev)
    charge = torch.zeros(1, device=dev)
    z_bad = g["numbers"].copy()
    z_bad[17] = 77
    mol_bad = np.zeros(96, dtype=np.int64)
    mol_bad[40] = 3
    for numbers, mol in ((z_bad, np.zeros(96, dtype=np.int64)), (g["numbers"], mol_bad)):
        msgs = []
        for v in (1, 0):
            hip_engine.set_option("prep_fused", v)
            try:
                with pytest.raises(ValueError) as ei:
                    hip_engine.eval(coord, torch.from_numpy(numbers).to(dev), torch.from_numpy(mol).to(dev), charge, cell=cell, forces=True,
                                    coulomb="dsf", dsf_rc=9.0)
                msgs.append(str(ei.value))
            finally:
                hip_engine.set_option("prep_fused", 1)
        assert msgs[0] == msgs[1]


@pytest.mark.parametrize("case", ["taxol", "batch5", "pbc2x96", "nse_batch5"])
def test_molecule_sums_inside_build_zbar_agree_with_the_partial_sum_launch(hip_engine, hip_engine_nse, case):
    """Option "nse_merged" (systems of up to 1 024 atoms): build_zbar's blocks form sum_i qbar_i f_i of their molecules themselves
    instead of reading the result of a launch in front.  The same terms in another association: everything to fp32 rounding of the
    evaluation, charges untouched (the adjoint only), and the merged form repeats bit for bit."""
    eng = hip_engine_nse if case == "nse_batch5" else hip_engine
    dev = eng.device
    kw = dict(forces=True, coulomb="simple")
    if case == "taxol":
        g = golden("taxol")
        c, z, mol, q = g["coord"], g["numbers"], np.zeros(113, dtype=np.int64), np.zeros(1, np.float32)
    elif case == "batch5":
        g = golden("batch5")
        c, z, mol, q = g["coord"], g["numbers"], g["mol_idx"], np.atleast_1d(g["charge"]).astype(np.float32)
    elif case == "pbc2x96":
        g = golden("pbc2x96_dsf9")
        c, z, mol, q = g["coord"], g["numbers"], g["mol_idx"], np.atleast_1d(g["charge"]).astype(np.float32)
        kw.update(cell=torch.from_numpy(g["cell"]).to(dev), stress=True, coulomb="dsf", dsf_rc=9.0)
    else:
        g = golden("nse")
        c, z, mol, q = g["b5_coord"], g["b5_numbers"], g["b5_mol_idx"], P._nse_charge(g["b5_charge"], g["b5_mult"])

    def go():
        r = eng.eval(torch.from_numpy(np.asarray(c, np.float32)).to(dev), torch.from_numpy(np.asarray(z)).to(dev),
                     torch.from_numpy(np.asarray(mol)).to(dev), torch.from_numpy(np.asarray(q, np.float32)).to(dev), **kw)
        return {k: v.cpu().numpy() for k, v in r.items()}

    a, b = both(eng, go, "nse_merged")
    a2 = go()
    for k in a:
        assert np.array_equal(a[k], a2[k]), k
    assert np.array_equal(a["charges"], b["charges"]) and np.array_equal(a["energy"], b["energy"])
    assert np.abs(a["forces"] - b["forces"]).max() < 2e-5 * max(1.0, np.abs(b["forces"]).max())
    if "stress" in a:
        assert np.abs(a["stress"] - b["stress"]).max() < 2e-6 * max(1e-2, np.abs(b["stress"]).max())


def test_d3_coordination_numbers_riding_on_the_matrix_build(hip_engine):
    """Option "d3_cn_rides": cn_i and the reference weights of DFT-D3 formed inside the cell-grid build of the D3 matrix instead of by a
    pass over it - the same terms, summed per lane in the builder's candidate order instead of the matrix' slot order."""
    g = golden("dftd3")
    par, tables = P._d3(12.0)
    hip_engine.set_dftd3_tables(tables)
    mol = np.zeros(96, dtype=np.int64)

    def go():
        return P._run_d3(hip_engine, g["pbc_coord"], g["pbc_numbers"], mol, "dsf", par, cell=g["pbc_cell"], stress=True, dsf_rc=9.0)[0]

    a, b = both(hip_engine, go, "d3_cn_rides")
    assert abs(a["energy"][0] - b["energy"][0]) < 1e-6
    assert np.abs(a["forces"] - b["forces"]).max() < 1e-5 * max(1.0, np.abs(b["forces"]).max())
    assert np.abs(a["stress"] - b["stress"]).max() < 1e-7
    assert np.array_equal(a["charges"], b["charges"])
