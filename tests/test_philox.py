"""CPU: Philox4x32-10 known-answer vectors (Random123 kat_vectors) for BOTH implementations:
the oracle's (oracle/uavenv_oracle.c) and the product's (csrc/philox.h via the C ABI host entry)."""
import pytest

KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


@pytest.mark.parametrize("ctr,key,want", KAT)
def test_oracle_philox_kat(ctr, key, want):
    from oracle import oracle as O

    assert tuple(O.philox4x32_10(ctr, key)) == want


@pytest.mark.parametrize("ctr,key,want", KAT)
def test_product_philox_kat(ctr, key, want):
    from drl_uav_cellularnet_amd import _capi, build

    build.build()
    assert tuple(_capi.philox4x32_10(ctr, key)) == want
