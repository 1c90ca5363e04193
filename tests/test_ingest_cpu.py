"""Text format of the reference's dumps (write_matrix / write_vector / read_matrix, src/tests/test_utils.f90:118-166):
the engine's parser, on the host, against Python's own correctly rounded float() - bit exact."""
import numpy as np
import pytest

from fortran_davidson_amd.engine_c import DavidsonHipError, parse_text_f64


def fortran_list_directed(values, width=25):
    # flang/gfortran list-directed output of a real(dp): leading blanks, 17 significant digits, E exponent
    return "".join(" %*.16E     \n" % (width, v) for v in values).encode()


def test_parser_is_bit_exact_on_write_matrix_style_dumps(golden):
    _, arrays = golden
    A = arrays["matrix_txt__A"]                     # the reference's own 100 x 100 test matrix
    text = fortran_list_directed(A.reshape(-1))     # row-major, one value per line
    out = parse_text_f64(text)
    assert out.size == A.size
    assert np.array_equal(out.reshape(A.shape), A)


def test_parser_matches_python_float_over_the_double_range():
    rng = np.random.default_rng(5)
    v = rng.standard_normal(50000) * 10.0 ** rng.integers(-300, 300, 50000)
    v[:6] = [0.0, -0.0, 5e-324, 1.7976931348623157e308, 2.2250738585072014e-308, 1.0]
    for fmt in ("%.17g", "%25.17E", "%.16e", "%.3f"):
        toks = [fmt % x for x in v]
        out = parse_text_f64(("\n".join(toks) + "\n").encode())
        assert np.array_equal(out, np.array([float(t) for t in toks]))


def test_parser_accepts_fortran_spellings():
    text = b" 1.0D+00, -2.5d-3\n3*0.5 +7 1.0+05 -1.5-03 \r\n\t1.0000000000000000     \n2*-1.E0,.5"
    out = parse_text_f64(text)
    assert out.tolist() == [1.0, -2.5e-3, 0.5, 0.5, 0.5, 7.0, 1.0e5, -1.5e-3, 1.0, -1.0, -1.0, 0.5]
    assert parse_text_f64(b"").size == 0
    assert parse_text_f64(b" \n\r\n ,").size == 0


@pytest.mark.parametrize("bad", [b"1.0 abc 2.0", b"1.0e", b"--1", b"0*1.0", b"1.2.3"])
def test_parser_rejects_garbage_loudly(bad):
    with pytest.raises(DavidsonHipError, match="not a number"):
        parse_text_f64(bad)


def test_parallel_parse_of_a_large_buffer_keeps_order():
    v = np.arange(400000, dtype=np.float64) * 0.125 - 1000.0
    text = fortran_list_directed(v)
    assert len(text) > (1 << 20)                    # large enough for the threaded path
    assert np.array_equal(parse_text_f64(text), v)
