"""Curve constants a caller of the engine needs on the host side (byte forms as the C ABI takes them: 64-byte affine points x || y,
big-endian).  Values: SEC 2 for secp256k1 (the reference holds them in src/group_impl.h:38-47,72 as SECP256K1_G / secp256k1_ge_const_g and in
src/scalar_impl.h as the group order); `secp256k1_generator_h` from the reference's src/modules/generator/main_impl.h:30-37."""
P = 2**256 - 2**32 - 977
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
G_XY = bytes.fromhex("79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798"
                     "483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8")
GENERATOR_H = bytes.fromhex("50929b74c1a04954b78b4b6035e97a5e078a5a0f28ec96d547bfee9ace803ac0"
                            "31d3c6863973926e049e637cb1b5f40a36dac28af1766968c30c2313f3a38904")
