// libm_exact.h -- sin / cos ... as glibc 2.35 computes them on an x86-64 machine with FMA, bit for bit.
//
// Why: the reference's CPU path calls glibc; wherever a transcendental feeds a decision that amplifies one ulp (fisheye /
// panorama primary rays -> hit position -> the hierarchical edge pick, which rescales ONE random number ~100 times,
// /root/reference/src/camera.h:142-191, /root/reference/src/edge.cpp:1115-1237) the device's own libm (ocml), which differs
// from glibc in the last bit of a few per cent of its results, draws other -- equally valid -- samples, and sample-exact
// parity with the oracle ends.  These routines restate the algorithms glibc 2.35 publishes (IBM Accurate Mathematical
// Library, sysdeps/ieee754/dbl-64/s_sin.c, e_atan2.c, s_atan.c, e_asin.c; Szabolcs Nagy's e_log.c / e_pow.c) with the
// fused multiply-adds exactly where that library's x86-64 `_fma` build has them (glibc selects that build at load time on
// every machine with FMA + AVX2: the oracle's hosts); all other operations are IEEE + - * / sqrt, which gfx950 rounds as
// the CPU does (the build has no contraction: -ffp-contract=off, `fma()` is written where a fused operation is meant).
// tests/test_libm_exact.py holds them to glibc bit for bit on 10^7 ... 10^8 arguments per function on the CPU and runs the same
// arguments through a kernel on the GPU.
//
// Range: |x| < 105414350 for sin / cos (beyond that glibc reduces with a 1200-bit 2/pi table; a renderer's angles never
// get there: those arguments go to the platform's own routine).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace gm {

RDR_FN double bits2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
RDR_FN uint64_t d2bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
RDR_FN uint32_t hi_word(double d) { return (uint32_t)(d2bits(d) >> 32); }
RDR_FN uint32_t lo_word(double d) { return (uint32_t)d2bits(d); }

// sin(k / 128), cos(k / 128), k = 0 ... 109, each as a (value, tail) pair -- glibc's __sincostab (sincostab.c)
RDR_FN const double *sincos_table() {
    static const double tab[440] = {
        0x0.0p+0, 0x0.0p+0, 0x1.0000000000000p+0, 0x0.0p+0,
        0x1.fffeaaaaeeeefp-8, -0x1.e45e2ec67b77cp-62, 0x1.fffc000155552p-1, 0x1.f4a01a0196daep-55,
        0x1.fffaaaaeeeed5p-7, -0x1.2ab639a9f0777p-63, 0x1.fff000155549fp-1, 0x1.28a28a03a5ef3p-55,
        0x1.7ff7001033255p-6, 0x1.efe2b51527336p-64, 0x1.ffdc006bff7e6p-1, 0x1.ae6dae86977bdp-55,
        0x1.ffeaaaeeee86fp-6, -0x1.cd406fb224ae2p-60, 0x1.ffc00155527d3p-1, -0x1.3b54492d89b5bp-55,
        0x1.3feb2b12d45d5p-5, 0x1.4ec54203d1c11p-60, 0x1.ff9c03414a7bap-1, 0x1.991f4be6c59bfp-57,
        0x1.7fdc01032fba9p-5, -0x1.599bdf46e997ap-59, 0x1.ff7006bfdf99fp-1, -0x1.8b3b560648d5fp-56,
        0x1.bfc6d78586dacp-5, 0x1.8e4fd03dbf236p-62, 0x1.ff3c0c8103a31p-1, 0x1.4856dbddc0e66p-56,
        0x1.ffaaaeeed4edbp-5, -0x1.2d16d32684b69p-59, 0x1.ff0015549f4d3p-1, 0x1.328387b99426fp-55,
        0x1.1fc343d808befp-4, -0x1.f3d32e6f3be4fp-58, 0x1.febc222a8ef9fp-1, 0x1.7934934f54c77p-58,
        0x1.3facb12d1755bp-4, -0x1.921915299468cp-58, 0x1.fe7034129ef6fp-1, -0x1.cbf4337c96f97p-57,
        0x1.5f911fd10b737p-4, -0x1.0184f02be9102p-58, 0x1.fe1c4c3c873ebp-1, -0x1.5a9c9057c4a02p-60,
        0x1.7f701032550e4p-4, 0x1.afc2d1800501ap-60, 0x1.fdc06bf7e6b9bp-1, 0x1.31902b535f8dbp-55,
        0x1.9f4902d55d1f9p-4, 0x1.2696d7eac1dc1p-58, 0x1.fd5c94b43e000p-1, -0x1.2e768cb4f92f9p-57,
        0x1.bf1b78568391dp-4, 0x1.e91841dea4cc8p-58, 0x1.fcf0c800e99b1p-1, 0x1.ea3d786d186acp-57,
        0x1.dee6f16c1cce6p-4, -0x1.50f8e2fb71673p-59, 0x1.fc7d078d1bc88p-1, 0x1.075d2447db685p-55,
        0x1.feaaeee86ee36p-4, -0x1.afcb2bcc6f03bp-59, 0x1.fc015527d5bd3p-1, 0x1.b68f35094efb8p-55,
        0x1.0f3378ddd71d1p-3, 0x1.d8468724f0f9ep-57, 0x1.fb7db2bfe0695p-1, 0x1.21dadf4f65ab1p-55,
        0x1.1f0d3d7afceafp-3, -0x1.6ef95099769a5p-57, 0x1.faf22263c4bd3p-1, -0x1.52ace133a2769p-58,
        0x1.2ee285e4ab88fp-3, -0x1.e4d0f05dee058p-57, 0x1.fa5ea641c36f2p-1, 0x1.04da6ed17cc7cp-59,
        0x1.3eb312c5d66cbp-3, 0x1.47d666b66cb91p-57, 0x1.f9c340a7cc428p-1, 0x1.c5b6b063b7462p-55,
        0x1.4e7ea4dc5f27bp-3, 0x1.949db2ac072fcp-58, 0x1.f91ff40374d01p-1, -0x1.7d03f4d3a9e4cp-57,
        0x1.5e44fcfa126f3p-3, -0x1.6f443063f89b6p-57, 0x1.f874c2e1eecf6p-1, -0x1.c6514e1332b16p-55,
        0x1.6e05dc05a4d4cp-3, -0x1.32c5c8b81c940p-66, 0x1.f7c1afeffde24p-1, -0x1.8f55bc47540b1p-56,
        0x1.7dc102fbaf2b5p-3, 0x1.5ab50e23c97c3p-59, 0x1.f706bdf9ece1cp-1, -0x1.698c80c36dcb4p-55,
        0x1.8d7632efaa944p-3, -0x1.20fa262cbb953p-57, 0x1.f643efeb82acdp-1, 0x1.6b00ac1fe28acp-56,
        0x1.9d252d0cec312p-3, 0x1.9c43d80b1137dp-58, 0x1.f57948cff6797p-1, 0x1.e3a0d3e03b1d5p-57,
        0x1.accdb297a0765p-3, -0x1.9883b57d6cdebp-58, 0x1.f4a6cbd1e3a79p-1, 0x1.13df0edaebb57p-55,
        0x1.bc6f84edc6199p-3, 0x1.9c1a56a7b0cabp-57, 0x1.f3cc7c3b3d16ep-1, -0x1.21a3ad28a3494p-57,
        0x1.cc0a6588289a3p-3, -0x1.868d09bc87c6bp-57, 0x1.f2ea5d753ffedp-1, 0x1.cc4215f56d583p-55,
        0x1.db9e15fb5a5d0p-3, -0x1.32e20d6cc6fc2p-57, 0x1.f20073086649fp-1, 0x1.b940416c1984bp-56,
        0x1.eb2a57f8ae5a3p-3, -0x1.0be06af572cebp-57, 0x1.f10ec09c5873bp-1, 0x1.d9072762c1283p-55,
        0x1.faaeed4f31577p-3, -0x1.15d88508e32b8p-57, 0x1.f01549f7deea1p-1, 0x1.d3c1e99e5cafdp-55,
        0x1.0515cbf65155cp-2, -0x1.9b8c29dfd8ec8p-56, 0x1.ef141300d2f26p-1, -0x1.2aa1b08ded372p-55,
        0x1.0cd00cef36436p-2, -0x1.9fb0a0c93e2b5p-56, 0x1.ee0b1fbc0f11cp-1, -0x1.bfd2380bbc3b1p-59,
        0x1.14861aa94ddebp-2, -0x1.be881b5b615a4p-57, 0x1.ecfa744d5efa1p-1, -0x1.56d0a4af541d0p-58,
        0x1.1c37d64c6b876p-2, 0x1.46076fe0dcff5p-56, 0x1.ebe214f76efa8p-1, -0x1.02f9f12ba543ep-55,
        0x1.23e52111aaf36p-2, -0x1.4f080334eff18p-56, 0x1.eac2061bbaf4fp-1, 0x1.2c1d53e94658dp-57,
        0x1.2b8ddc43eb49fp-2, 0x1.1553899f2d807p-57, 0x1.e99a4c3a7cd83p-1, -0x1.2264b1bc53ce8p-55,
        0x1.3331e94049f87p-2, 0x1.e0cb6b40c302cp-56, 0x1.e86aebf29a9edp-1, 0x1.9397afdbb58a7p-55,
        0x1.3ad129769d3d8p-2, 0x1.03d5504878398p-63, 0x1.e733ea0193d40p-1, -0x1.6428b3546ce13p-55,
        0x1.426b7e69ee697p-2, -0x1.f09c75705c59fp-56, 0x1.e5f54b436e9d0p-1, 0x1.7eb0fd02fc8bcp-55,
        0x1.4a00c9b0f3d20p-2, 0x1.823ba6bb08eadp-56, 0x1.e4af14b2a449cp-1, -0x1.68ca02e8a6833p-55,
        0x1.5190ecf68a77ap-2, 0x1.b357155eef0f3p-56, 0x1.e3614b680d6a5p-1, -0x1.27793aa015237p-56,
        0x1.591bc9fa2f597p-2, 0x1.7c74bac3fe0cbp-57, 0x1.e20bf49acd6c1p-1, -0x1.660aec7ef636cp-58,
        0x1.60a1429078775p-2, 0x1.b1fd80ba89133p-58, 0x1.e0af15a03dbcep-1, 0x1.fe8e702771ae6p-58,
        0x1.682138a38d7f7p-2, -0x1.d889202444aadp-56, 0x1.df4ab3ebd875ep-1, -0x1.e2d8a7e6736c4p-55,
        0x1.6f9b8e33a0255p-2, 0x1.42bc14ee9da0dp-56, 0x1.ddded50f228d6p-1, -0x1.e80c8d42ba2bfp-57,
        0x1.7710255764214p-2, -0x1.6ead7314bb6cep-57, 0x1.dc6b7eb995912p-1, 0x1.4b364776dcd35p-58,
        0x1.7e7ee03c86d4ep-2, -0x1.b63bcdabf5af2p-56, 0x1.daf0b6b888e83p-1, 0x1.a249e2b5e5ceap-55,
        0x1.85e7a12826949p-2, 0x1.8a40e9b5face0p-56, 0x1.d96e82f71a9dcp-1, 0x1.ff61bd5d2039dp-55,
        0x1.8d4a4a774992fp-2, 0x1.44a02ea766326p-56, 0x1.d7e4e97e17b4ap-1, -0x1.3b770352bed94p-57,
        0x1.94a6be9f546c5p-2, -0x1.69ce13e683f58p-56, 0x1.d653f073e4040p-1, -0x1.76236434bec37p-55,
        0x1.9bfce02e80510p-2, 0x1.09e39a320b0a4p-56, 0x1.d4bb9e1c619e0p-1, 0x1.f34bb77858f61p-55,
        0x1.a34c91cc50ccap-2, -0x1.a310e3b50cecdp-58, 0x1.d31bf8d8d7c06p-1, 0x1.e60dd3089cbddp-56,
        0x1.aa95b63a09277p-2, -0x1.6293eb13c0381p-57, 0x1.d1750727d94f0p-1, 0x1.0d52b1ec1a48ep-55,
        0x1.b1d8305321617p-2, -0x1.ae242cb99f519p-56, 0x1.cfc6cfa52ad9fp-1, 0x1.8b5b5508f2a0dp-55,
        0x1.b913e30dbac43p-2, -0x1.e38ad2f6c3ff1p-56, 0x1.ce115909a82e5p-1, 0x1.1f139bb31109ap-55,
        0x1.c048b17b140a3p-2, 0x1.19fe6757e9fa7p-57, 0x1.cc54aa2b2972ep-1, 0x1.4ee162ba83a98p-57,
        0x1.c7767ec7fd19ep-2, -0x1.eb14d1a3d5826p-58, 0x1.ca90c9fc67d0bp-1, -0x1.46a81485e3462p-57,
        0x1.ce9d2e3d4a51fp-2, -0x1.2fc8a12dae298p-57, 0x1.c8c5bf8ce1a84p-1, 0x1.ab3d1a1590123p-56,
        0x1.d5bca34047661p-2, 0x1.28a44a75fc29cp-56, 0x1.c6f39208be53bp-1, -0x1.741dbfbaadb42p-55,
        0x1.dcd4c15329c9ap-2, 0x1.0d4c6e171fd9ap-56, 0x1.c51a48b8b175ep-1, -0x1.1bbb43b9aa880p-57,
        0x1.e3e56c1582a69p-2, -0x1.0a4821099f88fp-58, 0x1.c339eb01ddd81p-1, -0x1.caaf5ee82c5c0p-55,
        0x1.eaee8744b05f0p-2, -0x1.789b43c9b027dp-58, 0x1.c1528065b7d50p-1, -0x1.892111312e828p-55,
        0x1.f1eff6bc4f97bp-2, 0x1.17212f8a7525cp-56, 0x1.bf641081e7536p-1, 0x1.b7bd71628a9a1p-55,
        0x1.f8e99e76abc97p-2, 0x1.9d950af2d00a3p-58, 0x1.bd6ea310294f5p-1, 0x1.31bbcc88c109dp-56,
        0x1.ffdb628d2f57ap-2, 0x1.f4a992e905b6ap-57, 0x1.bb723fe630f32p-1, 0x1.72bd2452d0a39p-56,
        0x1.0362939c69955p-1, -0x1.2d8cd78397b01p-55, 0x1.b96eeef58840ep-1, 0x1.45a3cc78fade0p-58,
        0x1.06d3686946e5bp-1, 0x1.3f5ae4538ff1bp-55, 0x1.b764b84b704c2p-1, -0x1.f5848c21b389bp-55,
        0x1.0a4021e9e1001p-1, -0x1.6f643a13914f6p-55, 0x1.b553a410c104ep-1, 0x1.8ff7947027a16p-58,
        0x1.0da8b26b5672ep-1, -0x1.a58def0bee909p-55, 0x1.b33bba89c8948p-1, 0x1.ea6a51d1f6ca9p-55,
        0x1.110d0c4b69c3bp-1, 0x1.d918998809981p-55, 0x1.b11d04162a4c6p-1, 0x1.1dd561efbc0c2p-56,
        0x1.146d21f8b7f82p-1, 0x1.bf9535e2739a8p-56, 0x1.aef78930bd275p-1, -0x1.f836279746f94p-56,
        0x1.17c8e5f2eedb0p-1, 0x1.35e57102e2488p-57, 0x1.accb526f69de5p-1, 0x1.8fb6a8dd6b6ccp-55,
        0x1.1b204acb02fddp-1, -0x1.f190c70cbb5ffp-58, 0x1.aa98688308913p-1, -0x1.b83d607cd5070p-63,
        0x1.1e7343236574cp-1, 0x1.22a3fa4f41d5ap-56, 0x1.a85ed4373e02dp-1, 0x1.9be06385ec792p-57,
        0x1.21c1c1b0394cfp-1, 0x1.e5b324b23aa31p-58, 0x1.a61e9e72586afp-1, 0x1.58330e2fd453fp-55,
        0x1.250bb93788bbbp-1, 0x1.ea3d02457bccep-56, 0x1.a3d7d0352bdcfp-1, -0x1.68dbaeca19669p-55,
        0x1.28511c917a067p-1, -0x1.01df1d9a16b70p-55, 0x1.a18a729aee445p-1, 0x1.95e25736c0358p-60,
        0x1.2b91dea88421ep-1, -0x1.fa371db216ab0p-55, 0x1.9f368ed912f85p-1, -0x1.1d200c5791606p-55,
        0x1.2ecdf279a3082p-1, 0x1.d3557e0e7e37ep-55, 0x1.9cdc2e3f25e5cp-1, 0x1.3f99112993f62p-55,
        0x1.32054b148bc4fp-1, 0x1.f6b42095a135bp-55, 0x1.9a7b5a36a6514p-1, 0x1.722cfcc9fa7a9p-55,
        0x1.3537db9be0367p-1, 0x1.b327e7af040f0p-57, 0x1.98141c42e1310p-1, 0x1.d1ff80488f08dp-55,
        0x1.386597456282bp-1, -0x1.10fada93b07a8p-56, 0x1.95a67e00cb1fdp-1, -0x1.0befda21f862dp-55,
        0x1.3b8e715a2840ap-1, -0x1.97653a7d2f07bp-56, 0x1.93328926d9e92p-1, -0x1.bb77003600cdap-55,
        0x1.3eb25d36cd53ap-1, -0x1.be570e1570fc0p-58, 0x1.90b84784ddaf7p-1, -0x1.0feb10ab93b87p-56,
        0x1.41d14e4ba6790p-1, 0x1.4608fd287ecf5p-55, 0x1.8e37c303d9ad1p-1, -0x1.463a4b53d4bf8p-57,
        0x1.44eb381cf386bp-1, -0x1.3ed6c1e6a5505p-55, 0x1.8bb105a5dc900p-1, 0x1.863e03e9474c1p-55,
        0x1.48000e431159fp-1, -0x1.b194a7463ed10p-55, 0x1.89241985d871fp-1, 0x1.c48d9c413ed84p-55,
        0x1.4b0fc46aab761p-1, 0x1.0da05738cc59ap-61, 0x1.869108d77a6c6p-1, 0x1.338ffe2bfe9ddp-56,
        0x1.4e1a4e54ed51bp-1, -0x1.a492f89b7c76ap-55, 0x1.83f7dde701ca0p-1, -0x1.152cf609bc6e8p-59,
        0x1.511f9fd7b351cp-1, -0x1.5c0e861c48831p-55, 0x1.8158a31916d5dp-1, -0x1.de8b90b8228dep-57,
        0x1.541facddbb724p-1, 0x1.232c28520d391p-56, 0x1.7eb362eaa1488p-1, 0x1.a1d65a4a5959fp-58,
        0x1.571a6966d59b3p-1, 0x1.c843b4d0fb198p-58, 0x1.7c0827f09e54fp-1, -0x1.c73d6d72aee68p-57,
        0x1.5a0fc98813a12p-1, -0x1.d82e2b7d4227bp-55, 0x1.7956fcd7f6543p-1, -0x1.ab276e9d45ae4p-55,
        0x1.5cffc16bf8f0dp-1, 0x1.96cb370eb578ap-55, 0x1.769fec655211fp-1, -0x1.827d5cf8c68c5p-57,
        0x1.5fea4552a9e57p-1, 0x1.0b6cef7ee20b7p-55, 0x1.73e30174efba1p-1, -0x1.5d3ae3d94ad5fp-57,
        0x1.62cf49921ac79p-1, -0x1.edd9855b6241ap-55, 0x1.712046fa77678p-1, 0x1.425b0a5029c81p-55,
        0x1.65aec2963e755p-1, 0x1.126f96b71053cp-55, 0x1.6e57c800cf55ep-1, 0x1.60286dedbd0a6p-55,
        0x1.6888a4e134b2fp-1, -0x1.6b7d37644d5e6p-55, 0x1.6b898fa9efb5dp-1, 0x1.15ac786ccf4b2p-56,
        0x1.6b5ce50b7821ap-1, -0x1.5d5158f702e0fp-57, 0x1.68b5a92eb6253p-1, -0x1.9a91ad985f89cp-55,
        0x1.6e2b77c40bde1p-1, -0x1.0e729857fad53p-56, 0x1.65dc1fdeb8cbap-1, -0x1.97c1b47337c77p-58,
        0x1.70f451d0a8c40p-1, 0x1.97ede3885770dp-57, 0x1.62fcff20191c7p-1, 0x1.d9143895756efp-57,
        0x1.73b7680dea578p-1, -0x1.2248306dc12a2p-56, 0x1.6018526f563dfp-1, 0x1.46ca5e0e432d0p-55,
        0x1.7674af6f7b524p-1, 0x1.e9d3f94ac84a8p-56, 0x1.5d2e255f1f17ap-1, 0x1.0314104c8892bp-55,
        0x1.792c1d0041d52p-1, -0x1.abf05eeb354ebp-55, 0x1.5a3e839824077p-1, 0x1.428aa2759be62p-55,
        0x1.7bdda5e28b3c2p-1, 0x1.ad1197ccd0393p-59, 0x1.574978d8e83f2p-1, 0x1.f4714af282d23p-55,
        0x1.7e893f5037959p-1, 0x1.0eefbaa650c4cp-55, 0x1.544f10f592ca5p-1, -0x1.e7ae8e6c7a62fp-55,
        0x1.812ede9ae4ba4p-1, -0x1.7830adf402ddap-55, 0x1.514f57d7bf3dap-1, 0x1.47a108073c259p-56,
    };
    return tab;
}

namespace detail {
// s_sin.c: TAYLOR_SIN -- |x| < 0.126
RDR_FN double taylor_sin(double xx, double x, double dx) {
    const double s1 = -0x1.5555555555555p-3, s2 = 0x1.1111111110ecep-7, s3 = -0x1.a01a019db08b8p-13,
                 s4 = 0x1.71de27b9a7ed9p-19, s5 = -0x1.addffc2fcdf59p-26;
    double p = fma(xx, s5, s4);
    p = fma(xx, p, s3);
    p = fma(xx, p, s2);
    p = fma(xx, p, s1);
    double t = fma(fma(p, x, -(0.5 * dx)), xx, dx);
    return x + t;
}
struct SinCosPoly { double s, c; const double *e; };
constexpr double kBig = 0x1.8p+45, kSn3 = -0x1.5555555555515p-3, kSn5 = 0x1.11110e829872fp-7,
                 kCs2 = 0.5, kCs4 = -0x1.5555555555535p-5, kCs6 = 0x1.6c16bedd9e239p-10;
// s_sin.c: do_sin -- sin(x + dx) from the table entry nearest |x| and two short series
RDR_FN double do_sin(double x, double dx) {
    const double ax = fabs(x);
    if (ax < 0.126) return taylor_sin(x * x, x, dx);
    if (x <= 0) dx = -dx;
    const double u = kBig + ax;
    const double xr = ax - (u - kBig);
    const double *e = sincos_table() + 4 * (int)lo_word(u);
    const double xx = xr * xr;
    const double s = xr + fma(xr * xx, fma(xx, kSn5, kSn3), dx);
    const double c = fma(xr, dx, xx * fma(xx, fma(xx, kCs6, kCs4), kCs2));
    const double sn = e[0], ssn = e[1], cs = e[2], ccs = e[3];
    const double cor = fma(s, cs, fma(-c, sn, fma(s, ccs, ssn)));
    return copysign(sn + cor, x);
}
// s_sin.c: do_cos
RDR_FN double do_cos(double x, double dx) {
    if (x < 0) dx = -dx;
    const double ax = fabs(x);
    const double u = kBig + ax;
    const double xr = ax - (u - kBig) + dx;
    const double *e = sincos_table() + 4 * (int)lo_word(u);
    const double xx = xr * xr;
    const double s = fma(xr * xx, fma(xx, kSn5, kSn3), xr);
    const double c = xx * fma(xx, fma(xx, kCs6, kCs4), kCs2);
    const double sn = e[0], ssn = e[1], cs = e[2], ccs = e[3];
    const double cor = fma(-s, sn, fma(-c, cs, fma(-s, ssn, ccs)));
    return cs + cor;
}
// s_sin.c: reduce_sincos -- x = n pi/2 + (a + da), 2.426 < |x| < 105414350
RDR_FN int reduce_sincos(double x, double &a, double &da) {
    const double hpinv = 0x1.45f306dc9c883p-1, toint = 0x1.8p+52, mp1 = 0x1.921fb58000000p+0, mp2 = -0x1.dde973c000000p-27,
                 pp3 = -0x1.cb3b398000000p-55, pp4 = -0x1.d747f23e32ed7p-83;
    const double t = fma(x, hpinv, toint);
    const double xn = t - toint;
    const double y = fma(-xn, mp2, fma(-xn, mp1, x));
    const double t2 = fma(-xn, pp3, y);
    double db = fma(-xn, pp3, y - t2);
    const double b = fma(-xn, pp4, t2);
    db += fma(-xn, pp4, t2 - b);
    a = b; da = db;
    return (int)(lo_word(t) & 3);
}
RDR_FN double do_sincos(double a, double da, int n) {
    double r = (n & 1) ? do_cos(a, da) : do_sin(a, da);
    return (n & 2) ? -r : r;
}
constexpr double kHp0 = 0x1.921fb54442d18p+0, kHp1 = 0x1.1a62633145c07p-54;
} // namespace detail

// s_sin.c: __sin
RDR_FN double sin(double x) {
    using namespace detail;
    const uint32_t k = hi_word(x) & 0x7fffffffu;
    if (k < 0x3e500000u) return x;                            // |x| < 2^-26
    if (k < 0x3feb6000u) return do_sin(x, 0.0);               // |x| < 0.855469
    if (k < 0x400368fdu) return copysign(do_cos(kHp0 - fabs(x), kHp1), x);   // |x| < 2.426265
    if (k < 0x419921fbu) { double a, da; int n = reduce_sincos(x, a, da); return do_sincos(a, da, n); }
    return ::sin(x);
}
// s_sin.c: __cos
RDR_FN double cos(double x) {
    using namespace detail;
    const uint32_t k = hi_word(x) & 0x7fffffffu;
    if (k < 0x3e400000u) return 1.0;                          // |x| < 2^-27
    if (k < 0x3feb6000u) return do_cos(x, 0.0);
    if (k < 0x400368fdu) {
        const double y = kHp0 - fabs(x);
        const double a = y + kHp1;
        const double da = (y - a) + kHp1;
        return do_sin(a, da);
    }
    if (k < 0x419921fbu) { double a, da; int n = reduce_sincos(x, a, da); return do_sincos(a, da, n + 1); }
    return ::cos(x);
}

// atan(c) around the 241 points c = u_i ~ (16.25 + i) / 256: {c, atan(c), and the coefficients of the series of atan around c}
// -- glibc's cij (uatan.tbl)
RDR_FN const double *atan_table() {
    static const double tab[241 * 7] = {
        0x1.0400665e0244ep-4, 0x1.03a737b53dd20p-4, 0x1.fdf1fcf5cfb72p-1, -0x1.01eb3ce2ae4c2p-4, -0x1.4d29edd58a40dp-2, 0x1.fda4ad907a18ap-5, 0x1.814df4df65b18p-3,
        0x1.0fffdb9b88cd8p-4, 0x1.0f99c63645300p-4, 0x1.fdc08a3ded30fp-1, -0x1.0d9dc669c1aedp-4, -0x1.4c669f7138de2p-2, 0x1.0a12f29d085a7p-4, 0x1.7f0eecfd48d20p-3,
        0x1.1fff15a73d4f1p-4, 0x1.1f85f2bee2040p-4, 0x1.fd7b342b56d31p-1, -0x1.1d2b7b69dea40p-4, -0x1.4b5523922ecc9p-2, 0x1.18f93522b1a04p-4, 0x1.7bead5660f061p-3,
        0x1.2fffdb2524aa2p-4, 0x1.2f716e71790a0p-4, 0x1.fd31f53b496a4p-1, -0x1.2cad84aab7374p-4, -0x1.4a34b58dd2fb2p-2, 0x1.27c0ad0cecc18p-4, 0x1.789d25d2743d7p-3,
        0x1.3fffe0573f3acp-4, 0x1.3f59d1702f6a0p-4, 0x1.fce4db071acc2p-1, -0x1.3c20f64db3686p-4, -0x1.49059eb3bfe93p-2, 0x1.36659caf74fedp-4, 0x1.752691c011fb0p-3,
        0x1.4ffef894384d6p-4, 0x1.4f3ed0ce204c0p-4, 0x1.fc93ea8ea5a01p-1, -0x1.4b84f7b5457c9p-4, -0x1.47c807401f2f9p-2, 0x1.44e64b4f67209p-4, 0x1.7187d4c540b77p-3,
        0x1.5fff8df406528p-4, 0x1.5f22b3c73d820p-4, 0x1.fc3f1b1f60f13p-1, -0x1.5adb2cb7fa73bp-4, -0x1.467be2b1eb555p-2, 0x1.5343599edc463p-4, 0x1.6dc1b238f5059p-3,
        0x1.7000f8c4f0d56p-4, 0x1.6f04b495a2fa0p-4, 0x1.fbe67340dce97p-1, -0x1.6a2244d98e1adp-4, -0x1.4521614064df1p-2, 0x1.617aa2ba78a66p-4, 0x1.69d4f50a3d7acp-3,
        0x1.8000fbb4057cfp-4, 0x1.7ee27be2cd3a0p-4, 0x1.fb8a039ec9246p-1, -0x1.7957731d9c773p-4, -0x1.43b8db6dc7d72p-2, 0x1.6f88ad69547dfp-4, 0x1.65c26f633ce8cp-3,
        0x1.8fff239cf2b7fp-4, 0x1.8ebb79f979e80p-4, 0x1.fb29d435506e1p-1, -0x1.8879a69b9cdb5p-4, -0x1.4242885feafa9p-2, 0x1.7d6bab6191a0ep-4, 0x1.618afa7cb8bb5p-3,
        0x1.9fff96e2f0772p-4, 0x1.9e93ad32a9480p-4, 0x1.fac5d04a3ec40p-1, -0x1.978c253f6ea97p-4, -0x1.40be3089c36f6p-2, 0x1.8b25c885aeb77p-4, 0x1.5d2f763cadce1p-3,
        0x1.b00026316b097p-4, 0x1.ae68cce24cc00p-4, 0x1.fa5e0938c5c66p-1, -0x1.a68c376f14e4bp-4, -0x1.3f2c31696cd7cp-2, 0x1.98b3b722a2cb4p-4, 0x1.58b0c9067ad62p-3,
        0x1.c0008604f58b1p-4, 0x1.be3a705650780p-4, 0x1.f9f285a7a2773p-1, -0x1.b578f3d5ac0a4p-4, -0x1.3d8cbf767119fp-2, 0x1.a613dc7e31b88p-4, 0x1.540fdf5594565p-3,
        0x1.d00026cca4ebap-4, 0x1.ce07ec1298a80p-4, 0x1.f9834e8d36c4ap-1, -0x1.c45135bcac5fep-4, -0x1.3be018b5236f1p-2, 0x1.b34472e991970p-4, 0x1.4f4dab8adb373p-3,
        0x1.dfff4b2b47fcap-4, 0x1.ddd164a051d80p-4, 0x1.f910678dcc895p-1, -0x1.d3149f0966844p-4, -0x1.3a266744f9a5fp-2, 0x1.c0446edb7f27ap-4, 0x1.4a6b2583f9ecap-3,
        0x1.f000aa9a05be0p-4, 0x1.ed996a3bda540p-4, 0x1.f899c1b8ba97fp-1, -0x1.e1c512287a677p-4, -0x1.385f8edc130bbp-2, 0x1.cd14bf306ff50p-4, 0x1.45694a667a72bp-3,
        0x1.ffffaba8f63dep-4, 0x1.fd5b569fe4780p-4, 0x1.f81f84863dc7dp-1, -0x1.f05dbd1518706p-4, -0x1.368c44687a69cp-2, 0x1.d9b081b3868dap-4, 0x1.40491c345adfcp-3,
        0x1.07ffa6eccada8p-3, 0x1.068d00a396400p-3, 0x1.f7a19f1fcfc6bp-1, -0x1.fee0c861df0dfp-4, -0x1.34ac65a586c0cp-2, 0x1.e618f189d637ap-4, 0x1.3b0ba195779d4p-3,
        0x1.1000333432713p-3, 0x1.0e6b0f203d1a0p-3, 0x1.f7200fe0eb463p-1, -0x1.06a72e15cb19ap-3, -0x1.32c00b8db761ep-2, 0x1.f24d8a11f5e3ep-4, 0x1.35b1e569e85ddp-3,
        0x1.17ffcda1c4811p-3, 0x1.1646229ebda00p-3, 0x1.f69af7d558737p-1, -0x1.0dd170b33969bp-3, -0x1.30c7d33ac50d1p-2, 0x1.fe4aa9be43f0fp-4, 0x1.303cf692539cbp-3,
        0x1.1ffff3cca418dp-3, 0x1.1e1fa3b978ea0p-3, 0x1.f612445d421a9p-1, -0x1.14f03acac8aa8p-3, -0x1.2ec3962e675a3p-2, 0x1.0508c2fa6b426p-3, 0x1.2aade780a6467p-3,
        0x1.27ff7d9c78922p-3, 0x1.25f661b91e640p-3, 0x1.f5860f52e192cp-1, -0x1.1c023e5de2394p-3, -0x1.2cb3d6bee0abdp-2, 0x1.0acfb5e075c1ap-3, 0x1.2505cdffe453ap-3,
        0x1.2fff7a1fc1aaap-3, 0x1.2dcb583257c40p-3, 0x1.f4f64c719b6fbp-1, -0x1.2308261514083p-3, -0x1.2a9887f7b72d5p-2, 0x1.107a77c887402p-3, 0x1.1f45c2c3cd6d1p-3,
        0x1.380059d78e15ep-3, 0x1.359ee6ac98ee0p-3, 0x1.f462f944cec16p-1, -0x1.2a020d85b87a9p-3, -0x1.2871c2e4ab369p-2, 0x1.1608dc31a65d9p-3, 0x1.196ee130bbe50p-3,
        0x1.400049f431b1ap-3, 0x1.3d6f36bd65360p-3, 0x1.f3cc3dd99b68ap-1, -0x1.30ee1b3dd00edp-3, -0x1.26403f8482664p-2, 0x1.1b792fe136626p-3, 0x1.138246eac7440p-3,
        0x1.48004e01d95a1p-3, 0x1.453d386f00cc0p-3, 0x1.f3320e3970539p-1, -0x1.37ccf0a5279aap-3, -0x1.2403f3b151d5dp-2, 0x1.20cbbe331c9e6p-3, 0x1.0d81139e3f097p-3,
        0x1.4fff7aa9382ddp-3, 0x1.4d07f8c590a80p-3, 0x1.f294834df28e0p-1, -0x1.3e9d85b43915cp-3, -0x1.21bd5eb8845a2p-2, 0x1.25ff8ac6ac8adp-3, 0x1.076c688ed96cap-3,
        0x1.58006352408bep-3, 0x1.54d1ec39a73e0p-3, 0x1.f1f3709ae009cp-1, -0x1.4561cb9be8550p-3, -0x1.1f6c00053f52ep-2, 0x1.2b15def783be9p-3, 0x1.014568615239bp-3,
        0x1.5ffff2b193f81p-3, 0x1.5c9804f73e000p-3, 0x1.f14f1ae110e29p-1, -0x1.4c16e9098b3d2p-3, -0x1.1d10f8f058241p-2, 0x1.300c6a14fa897p-3, 0x1.f61a6d56607c0p-4,
        0x1.680084460e6e1p-3, 0x1.645c804a55e20p-3, 0x1.f0a758fa36ec5p-1, -0x1.52be9d62fa883p-3, -0x1.1aabd69a74048p-2, 0x1.34e451679eb02p-3, 0x1.e989ef7c14c3dp-4,
        0x1.6fffb9e99a846p-3, 0x1.6c1d04b35fd40p-3, 0x1.effc63ef8ef95p-1, -0x1.5956b76a2fe63p-3, -0x1.183d8ddc78ddfp-2, 0x1.399bdac606d66p-3, 0x1.dcdba070d286ap-4,
        0x1.780080ffcd490p-3, 0x1.73dc5b55758e0p-3, 0x1.ef4e0457e2065p-1, -0x1.5fe167d6ff9bcp-3, -0x1.15c579fadd384p-2, 0x1.3e34773e52d32p-3, 0x1.d011c9a65ae4bp-4,
        0x1.80006148e79c1p-3, 0x1.7b9812b7f8ca0p-3, 0x1.ee9c7701687edp-1, -0x1.665c70e1ef36dp-3, -0x1.13449ccbcbdabp-2, 0x1.42ac75c71b3e8p-3, 0x1.c32eb3e81980ep-4,
        0x1.880060f487c17p-3, 0x1.83511bc0e3640p-3, 0x1.ede7ad2d55329p-1, -0x1.6cc8737e644bap-3, -0x1.10bae60597557p-2, 0x1.4704313e26fbep-3, 0x1.b634a6fb18bf4p-4,
        0x1.90004d3518d76p-3, 0x1.8b0738874c100p-3, 0x1.ed2fb2ed6673bp-1, -0x1.732512a6ebac3p-3, -0x1.0e28a6924232fp-2, 0x1.4b3b573bcc03fp-3, 0x1.a925e8c72507fp-4,
        0x1.97fffd2f20d5cp-3, 0x1.92ba351af5920p-3, 0x1.ec7493d32449fp-1, -0x1.7971fc308255fp-3, -0x1.0b8e2d572d28fp-2, 0x1.4f51a337448fep-3, 0x1.9c04bcfcbc620p-4,
        0x1.a0005bf80f060p-3, 0x1.9a6ae6e9e8960p-3, 0x1.ebb641ef200e7p-1, -0x1.7fafb6e96e5c1p-3, -0x1.08eb6ec6ad647p-2, 0x1.53475f53d0ba6p-3, 0x1.8ed364433c20ep-4,
        0x1.a7ff7deeca8e4p-3, 0x1.a2176948578e0p-3, 0x1.eaf4f328ff98bp-1, -0x1.85dc958149b1cp-3, -0x1.06414f933a1abp-2, 0x1.571b760c45a8fp-3, 0x1.81941be58c308p-4,
        0x1.affff7defd553p-3, 0x1.a9c229eba6b80p-3, 0x1.ea30710a85e10p-1, -0x1.8bfa67f9dea61p-3, -0x1.038f35a474e8fp-2, 0x1.5acf030c225d2p-3, 0x1.74491d062812fp-4,
        0x1.b7ffe669932a5p-3, 0x1.b1694cff6dfe0p-3, 0x1.e968f1921d387p-1, -0x1.92078e075d95ap-3, -0x1.00d60526793c4p-2, 0x1.5e61073842a52p-3, 0x1.66f49c5331d5ap-4,
        0x1.bfff9b44759f3p-3, 0x1.b90d15073a2a0p-3, 0x1.e89e756598313p-1, -0x1.98041cfb9203dp-3, -0x1.fc2bcbed91b37p-3, 0x1.61d196d4fc2fcp-3, 0x1.5998c9411537ep-4,
        0x1.c80075568f3ecp-3, 0x1.c0aec4a31dbe0p-3, 0x1.e7d0e18f270a8p-1, -0x1.9df0ef522b132p-3, -0x1.f69d42179c242p-3, 0x1.6521336646fcdp-3, 0x1.4c37cdc699095p-4,
        0x1.cfff8601a799fp-3, 0x1.c84b849db66a0p-3, 0x1.e7008a0ee780ep-1, -0x1.a3cbb3a403934p-3, -0x1.f102fd490be32p-3, 0x1.684ea037d4137p-3, 0x1.3ed3cd9ec855ap-4,
        0x1.d7ff97bbf1497p-3, 0x1.cfe5f1e008ce0p-3, 0x1.e62d2f04615c7p-1, -0x1.a996515aade2cp-3, -0x1.eb5b90b44b682p-3, 0x1.6b5af92ec8d57p-3, 0x1.316ee60d831aep-4,
        0x1.e000840209b20p-3, 0x1.d77ddb145a760p-3, 0x1.e556dbe1dfdf1p-1, -0x1.af5082186af0fp-3, -0x1.e5a799420489dp-3, 0x1.6e462454feb2cp-3, 0x1.240b2d2945a8cp-4,
        0x1.e8000c0ae943cp-3, 0x1.df1113ca10100p-3, 0x1.e47dd59e7308bp-1, -0x1.b4f889439f69fp-3, -0x1.dfe93798de600p-3, 0x1.710f58f267389p-3, 0x1.16aab1a8a373ep-4,
        0x1.f00036d532803p-3, 0x1.e6a17cb4e5c80p-3, 0x1.e3a1ee3d0f6c2p-1, -0x1.ba8fb6e31f768p-3, -0x1.da1f7e6a382e3p-3, 0x1.73b75b36ac4c0p-3, 0x1.094f7a3470b0ap-4,
        0x1.f7ffa48b8afc3p-3, 0x1.ee2dbe1654560p-3, 0x1.e2c3543f2ab37p-1, -0x1.c014f598207d6p-3, -0x1.d44bf1efe809ap-3, 0x1.763dc698a561ep-3, 0x1.f7f70a7cf78a3p-5,
        0x1.00002eb334faep-2, 0x1.f5b7b77ab25e0p-3, 0x1.e1e1d78a5c127p-1, -0x1.c5898c555d571p-3, -0x1.ce6d9b706cf86p-3, 0x1.78a350823f643p-3, 0x1.dd6190b9118e8p-5,
        0x1.03ffca8af86fep-2, 0x1.fd3cbb53a0c00p-3, 0x1.e0fdcfdcbac8bp-1, -0x1.caeb76c3246ffp-3, -0x1.c8870d6e19ad3p-3, 0x1.7ae73d2c48e91p-3, 0x1.c2e260510fdb0p-5,
        0x1.07ffcd38984b7p-2, 0x1.025f75732d4a0p-2, 0x1.e017049c17ab3p-1, -0x1.d03c29afe5028p-3, -0x1.c29719a2c1833p-3, 0x1.7d0a569041dcfp-3, 0x1.a87d3f497c653p-5,
        0x1.0bfff1ed2add7p-2, 0x1.061edcd7f7420p-2, 0x1.df2d8da96b750p-1, -0x1.d57b2c777881ep-3, -0x1.bc9ea8692b503p-3, 0x1.7f0c942abf9e7p-3, 0x1.8e35e04b42bb4p-5,
        0x1.10003a8515cdap-2, 0x1.09dc9027416a0p-2, 0x1.de41734899950p-1, -0x1.daa867983ede4p-3, -0x1.b69e3999706b6p-3, 0x1.80ee1b0f126dbp-3, 0x1.740fe17ee9babp-5,
        0x1.14001f3af9cc5p-2, 0x1.0d980b6e1aba0p-2, 0x1.dd52de0412681p-1, -0x1.dfc316863b28bp-3, -0x1.b0971c55b8d5ap-3, 0x1.82aeda6731aacp-3, 0x1.5a0ecc73bd8f0p-5,
        0x1.18003b6122509p-2, 0x1.1151daa1e67a0p-2, 0x1.dc61b2e0c1f32p-1, -0x1.e4cbeb9ba6b7ep-3, -0x1.aa88e90c2431cp-3, 0x1.844f48bcbda5ep-3, 0x1.4036150e585ffp-5,
        0x1.1bfffa6a2a153p-2, 0x1.15096e7a18dc0p-2, 0x1.db6e1e1218f3fp-1, -0x1.e9c219621d6a2p-3, -0x1.a475022627b04p-3, 0x1.85cf5ff8b908ep-3, 0x1.268919833c0d6p-5,
        0x1.1fffd2d345aafp-2, 0x1.18bf3053bf760p-2, 0x1.da780cc3acb29p-1, -0x1.eea622aa756aep-3, -0x1.9e5b347ed9793p-3, 0x1.872f887ab542ap-3, 0x1.0d0b2158e9e9ap-5,
        0x1.23ffcf14cf05ap-2, 0x1.1c7324d568460p-2, 0x1.d97f855f32d3dp-1, -0x1.f378021d457c8p-3, -0x1.983bef065b845p-3, 0x1.886fffba70cd8p-3, 0x1.e77ebaeb85cccp-6,
        0x1.27ffe0bae6fc9p-2, 0x1.202539a27c160p-2, 0x1.d88494619176ep-1, -0x1.f83795c0ac9ecp-3, -0x1.9217c5e645195p-3, 0x1.8990ff4264515p-3, 0x1.b551ce6b92e65p-6,
        0x1.2c001a297a7dep-2, 0x1.23d57acb927c0p-2, 0x1.d7873e4958fb6p-1, -0x1.fce4e43572249p-3, -0x1.8bef19f3560f3p-3, 0x1.8a92cdf7f0e5bp-3, 0x1.83958116f3b19p-6,
        0x1.2fffe7267616ap-2, 0x1.27835b2f378c0p-2, 0x1.d687b13906586p-1, -0x1.00bf9afda1a0fp-2, -0x1.85c34c197ad7dp-3, 0x1.8b7591e99f0a7p-3, 0x1.524fa6525c365p-6,
        0x1.33ffe48153b20p-2, 0x1.2b2f66a2fdcc0p-2, 0x1.d585cf827fbe4p-1, -0x1.03039b45a6918p-2, -0x1.7f93e5dfc3f72p-3, 0x1.8c39bc5210022p-3, 0x1.2185e168fb62ep-6,
        0x1.380038122579ap-2, 0x1.2ed9baf6ec1e0p-2, 0x1.d4819872f20d3p-1, -0x1.053e81f4c1031p-2, -0x1.79612621ffd79p-3, 0x1.8cdf9db9d9dfcp-3, 0x1.e27b480c6852fp-7,
        0x1.3c0033ef39141p-2, 0x1.3281b4668c700p-2, 0x1.d37b418590d1ap-1, -0x1.076fea3ef2560p-2, -0x1.732c93033287ap-3, 0x1.8d676ca2e5458p-3, 0x1.82f85d80944b1p-7,
        0x1.4000163fa0e31p-2, 0x1.362787b565000p-2, 0x1.d272c47a813dap-1, -0x1.0997f493b9d88p-2, -0x1.6cf643da9fe3cp-3, 0x1.8dd18c1cd3331p-3, 0x1.248d1f70f6e07p-7,
        0x1.4400374071092p-2, 0x1.39cb80f0a4000p-2, 0x1.d16813ba47a6bp-1, -0x1.0bb6cd8788947p-2, -0x1.66be2589596a6p-3, 0x1.8e1e5c9b3ec1ep-3, 0x1.8e868d20fab86p-8,
        0x1.48000c880f200p-2, 0x1.3d6d1deffb460p-2, 0x1.d05b5cadc576cp-1, -0x1.0dcc2a1d352c2p-2, -0x1.608583d7d2574p-3, 0x1.8e4e303208bc0p-3, 0x1.ac9096379e732p-9,
        0x1.4c0004d97d2cbp-2, 0x1.410cbf3a2e220p-2, 0x1.cf4c8bb7ed511p-1, -0x1.0fd8437766a49p-2, -0x1.5a4c25aabc13cp-3, 0x1.8e616c80dac4bp-3, 0x1.038aab04695c2p-11,
        0x1.4fffd9397539fp-2, 0x1.44aa206a7dec0p-2, 0x1.ce3bbcf479ddep-1, -0x1.11daf4d122984p-2, -0x1.5412eb1024df0p-3, 0x1.8e5871b2c560dp-3, -0x1.25da8951c088dp-9,
        0x1.53ffff304715fp-2, 0x1.4845a791f3900p-2, 0x1.cd28da45e0fd8p-1, -0x1.13d478d61f221p-2, -0x1.4dd98d3e9bb99p-3, 0x1.8e33a0f181507p-3, -0x1.43c33d08bd25cp-8,
        0x1.58002e88ea386p-2, 0x1.4bdf0f575d6c0p-2, 0x1.cc14002035609p-1, -0x1.15c4ab808071ep-2, -0x1.47a0eb2945fcfp-3, 0x1.8df35fc056447p-3, -0x1.f2011b00a45cdp-8,
        0x1.5bffd70f4d590p-2, 0x1.4f75d284d7ae0p-2, 0x1.cafd5f2de98b6p-1, -0x1.17ab4a2b42f42p-2, -0x1.416a51c285a92p-3, 0x1.8d982511d6c5ap-3, -0x1.4ecc177008605p-7,
        0x1.5fffdb70d6e53p-2, 0x1.530ab8e2ff500p-2, 0x1.c9e4c32d2429dp-1, -0x1.1988c35190681p-2, -0x1.3b34cbf748319p-3, 0x1.8d22498d3a613p-3, -0x1.a33d4aa295f9fp-7,
        0x1.63ffc5c7399e2p-2, 0x1.569d54f022e80p-2, 0x1.c8ca558dd180fp-1, -0x1.1b5ce1d701de4p-2, -0x1.35017a7806a5ap-3, 0x1.8c92456c01cf9p-3, -0x1.f64d9942059e1p-7,
        0x1.67ffd9a1ac7d2p-2, 0x1.5a2ddf50031e0p-2, 0x1.c7ae0ceff6debp-1, -0x1.1d27c7c8c245bp-2, -0x1.2ed05c6aa933fp-3, 0x1.8be87ddc5cf1fp-3, -0x1.23fb6d594386fp-6,
        0x1.6bffd6f7b9353p-2, 0x1.5dbc1b4e066c0p-2, 0x1.c6900456b591ap-1, -0x1.1ee95c2d6d0aap-2, -0x1.28a23b11086f7p-3, 0x1.8b256dde22d5ap-3, -0x1.4c19a489d85a4p-6,
        0x1.6fffbf02a83e4p-2, 0x1.614806a237dc0p-2, 0x1.c57044cc81773p-1, -0x1.20a1a4b9029cap-2, -0x1.2277789f5fb1cp-3, 0x1.8a4989b09e911p-3, -0x1.737ec130d419ap-6,
        0x1.73ffe128c213ap-2, 0x1.64d1e42499480p-2, 0x1.c44ec129c0d30p-1, -0x1.2250c83787259p-2, -0x1.1c4ffd55be4fcp-3, 0x1.8955336b2d603p-3, -0x1.9a2842e43df46p-6,
        0x1.77ffbea0cdc7ap-2, 0x1.6859405b0e220p-2, 0x1.c32ba687132c0p-1, -0x1.23f697273497ep-2, -0x1.162cecd39b037p-3, 0x1.8848ffa930aafp-3, -0x1.c013da4554412p-6,
        0x1.7c003f18edab8p-2, 0x1.6bdee4127bee0p-2, 0x1.c206bc01607bdp-1, -0x1.259375fee2f42p-2, -0x1.100d4307761e1p-3, 0x1.872525dfec556p-3, -0x1.e53f67958f973p-6,
        0x1.7fffd41f35c4cp-2, 0x1.6f616da6607a0p-2, 0x1.c0e07cddc8437p-1, -0x1.2726cbfb4daeap-2, -0x1.09f3be0db1472p-3, 0x1.85ea92a95aa1bp-3, -0x1.04d47d872cfa2p-5,
        0x1.8400326c7c46bp-2, 0x1.72e2596b8be00p-2, 0x1.bfb874cdedf38p-1, -0x1.28b14d09404f3p-2, -0x1.03de1e7fb61f2p-3, 0x1.84993acb33be9p-3, -0x1.16a769b1de607p-5,
        0x1.88003ca90b179p-2, 0x1.7660aa104a220p-2, 0x1.be8eff236e2f6p-1, -0x1.2a32919a94ddfp-2, -0x1.fb9ce0856a081p-4, 0x1.8331f33f70280p-3, -0x1.2817af01308ccp-5,
        0x1.8c003e9692fd5p-2, 0x1.79dc9f0b2cb00p-2, 0x1.bd640f2966495p-1, -0x1.2baabfd6ec2eap-2, -0x1.ef892e08e9c2dp-4, 0x1.81b52031873e3p-3, -0x1.39249ac12113dp-5,
        0x1.8fffe35be5c5fp-2, 0x1.7d55ebdccdfc0p-2, 0x1.bc37c6eabcf77p-1, -0x1.2d19c2d74f445p-2, -0x1.e382ce63f2cdbp-4, 0x1.802360e6fe2aep-3, -0x1.49cd90e66ab41p-5,
        0x1.94002aa8974cdp-2, 0x1.80cd6b8afd880p-2, 0x1.bb09e4468ccbap-1, -0x1.2e7ffec84e686p-2, -0x1.d787688c659e8p-4, 0x1.7e7ccc2f15460p-3, -0x1.5a120b410d3edp-5,
        0x1.98002e08efdeap-2, 0x1.8442534856920p-2, 0x1.b9dab3f290478p-1, -0x1.2fdd2bb81edefp-2, -0x1.cb9a531e68398p-4, 0x1.7cc23c2dbb11bp-3, -0x1.69f1998467e78p-5,
        0x1.9c00275294b6bp-2, 0x1.87b4d299f6200p-2, 0x1.b8aa2de96cf1fp-1, -0x1.313168c4d45d2p-2, -0x1.bfbb7edce4dbap-4, 0x1.7af418907fec9p-3, -0x1.796be07419f55p-5,
        0x1.a0002f3e490ecp-2, 0x1.8b24fc21a4500p-2, 0x1.b77853b5ef7ddp-1, -0x1.327cc8eae70cdp-2, -0x1.b3eb3d49e40dap-4, 0x1.7912d4d93f7eap-3, -0x1.888099e21606ap-5,
        0x1.a3fff458461b6p-2, 0x1.8e9287754d2c0p-2, 0x1.b64546a0daf0ep-1, -0x1.33bf3dc2a9a3fp-2, -0x1.a82b14917d003p-4, 0x1.771f17c7566cfp-3, -0x1.972f93d700dd8p-5,
        0x1.a800287e12aaep-2, 0x1.91fe0a5dfd000p-2, 0x1.b510da0d82e05p-1, -0x1.34f90a76ad312p-2, -0x1.9c798deec35adp-4, 0x1.751908a0ef43ep-3, -0x1.a578b0872efc8p-5,
        0x1.ac00149a86c84p-2, 0x1.9566e5c4516e0p-2, 0x1.b3db4dd03f6b6p-1, -0x1.362a0291c1f82p-2, -0x1.90d9503f6df60p-4, 0x1.7301825091e92p-3, -0x1.b35be577a022bp-5,
        0x1.affff2f4cc2e1p-2, 0x1.98cd494226540p-2, 0x1.b2a499297200ap-1, -0x1.375245153fd01p-2, -0x1.854a3ae3de27ep-4, 0x1.70d8e7eb3f331p-3, -0x1.c0d93b6ad570ep-5,
        0x1.b4000c2f3711ep-2, 0x1.9c31701cdc4c0p-2, 0x1.b16caea63781bp-1, -0x1.3871f3665b649p-2, -0x1.79cc03f70fbc6p-4, 0x1.6e9f9061dfc2ep-3, -0x1.cdf0cd837f9c3p-5,
        0x1.b8000a777e180p-2, 0x1.9f930f3748f20p-2, 0x1.b033b0fb0162ap-1, -0x1.3989025978cabp-2, -0x1.6e6025c765aabp-4, 0x1.6c5629c16d678p-3, -0x1.daa2c92a16ebfp-5,
        0x1.bbffd087e14edp-2, 0x1.a2f20bf0ddb00p-2, 0x1.aef9b1cce6e94p-1, -0x1.3a9778b73e3c3p-2, -0x1.6307709efd1ccp-4, 0x1.69fd458408d3ap-3, -0x1.e6ef6d2e48013p-5,
        0x1.c0000f0086783p-2, 0x1.a64ef8d448080p-2, 0x1.adbe835990b5ap-1, -0x1.3b9d927241b86p-2, -0x1.57c06c20e4001p-4, 0x1.6794f90e6c8abp-3, -0x1.f2d709a630a27p-5,
        0x1.c4001863e58f8p-2, 0x1.a9a941c3a1ba0p-2, 0x1.ac82635ed7dd2p-1, -0x1.3c9b30c075b50p-2, -0x1.4c8d7a429793cp-4, 0x1.651e295903c22p-3, -0x1.fe59ff0f8b649p-5,
        0x1.c7ffc6c62c3bfp-2, 0x1.ad00c580a5840p-2, 0x1.ab45662d1d808p-1, -0x1.3d905acbb06ecp-2, -0x1.416f7421e42dcp-4, 0x1.62996e5608efdp-3, -0x1.04bc5f14b649ap-4,
        0x1.cc00234b2a209p-2, 0x1.b0565f68f3b40p-2, 0x1.aa0741e3dc946p-1, -0x1.3e7d5e2db674ep-2, -0x1.3663ea4833ffep-4, 0x1.60069c4f0392bp-3, -0x1.0a19e38b10201p-4,
        0x1.cfffcaac5f9f9p-2, 0x1.b3a8e59c45cc0p-2, 0x1.a8c86d2389c24p-1, -0x1.3f61f8362b2cbp-2, -0x1.2b6f1c6c746a6p-4, 0x1.5d671426d2946p-3, -0x1.0f45d4981ce75p-4,
        0x1.d40040d800c64p-2, 0x1.b6f9988af6580p-2, 0x1.a78877498ced2p-1, -0x1.403e8ef8975c0p-2, -0x1.208d4bea81e2bp-4, 0x1.5aba5283ffa4ep-3, -0x1.1440811705130p-4,
        0x1.d7ffeb0e64500p-2, 0x1.ba4722324e140p-2, 0x1.a647e8c5ad680p-1, -0x1.4112da03f042dp-2, -0x1.15c339580389cp-4, 0x1.5801e49d9889ep-3, -0x1.190a3ef96554fp-4,
        0x1.dbffe2dfcf4ebp-2, 0x1.bd9269f1d27a0p-2, 0x1.a50671ac286cap-1, -0x1.41df2590a4de1p-2, -0x1.0b0e48bd1efa5p-4, 0x1.553d8702506d0p-3, -0x1.1da36ada415a6p-4,
        0x1.dfffd8a34bbc2p-2, 0x1.c0db2c4f7a2c0p-2, 0x1.a3c432ef70bb3p-1, -0x1.42a3716ee647cp-2, -0x1.006fadb6270bbp-4, 0x1.526de86f08de6p-3, -0x1.220c67e5061fbp-4,
        0x1.e3ffdd26415c0p-2, 0x1.c421758282940p-2, 0x1.a2812f391ddcbp-1, -0x1.435fd18eddf0ap-2, -0x1.ebcf288a589afp-5, 0x1.4f9374cf96163p-3, -0x1.26459f6a18481p-4,
        0x1.e7fff37f72672p-2, 0x1.c765467aa3dc0p-2, 0x1.a13d6d6ce86b3p-1, -0x1.4414574037e91p-2, -0x1.d6ec93b2cc445p-5, 0x1.4caea0564f101p-3, -0x1.2a4f80c49cd64p-4,
        0x1.ebffda11bc00fp-2, 0x1.caa6685e23660p-2, 0x1.9ff90a25c2396p-1, -0x1.44c108a64724fp-2, -0x1.c23992f871e82p-5, 0x1.49c010afbfb85p-3, -0x1.2e2a80f0ff3fep-4,
        0x1.effff3313756dp-2, 0x1.cde529d30cc20p-2, 0x1.9eb3edff9491fp-1, -0x1.456607e6abaaep-2, -0x1.adb4c3e8aa98dp-5, 0x1.46c7f25d8ff7dp-3, -0x1.31d71a71d448dp-4,
        0x1.f4001914b856ep-2, 0x1.d1216aac1bb20p-2, 0x1.9d6e2c9bc4315p-1, -0x1.46036004e7e91p-2, -0x1.995f7fb901f89p-5, 0x1.43c6d3f5be04ap-3, -0x1.3555cce8abf92p-4,
        0x1.f8003cd144428p-2, 0x1.d45b1d93e9640p-2, 0x1.9c27d256fdfebp-1, -0x1.4699209f7c145p-2, -0x1.853a9ed521174p-5, 0x1.40bd32b27751fp-3, -0x1.38a71cfa5c5f2p-4,
        0x1.fc00200545bd9p-2, 0x1.d7920f536d960p-2, 0x1.9ae0faae99ea5p-1, -0x1.4727538dd66f4p-2, -0x1.7147db5484f74p-5, 0x1.3dabaf8efc373p-3, -0x1.3bcb93ea6b864p-4,
        0x1.ffffbda6f2aa8p-2, 0x1.dac63b420faa0p-2, 0x1.9999aed4d0cabp-1, -0x1.47ae0bfcc6072p-2, -0x1.5d87c25bf7a4ap-5, 0x1.3a92bf5999ee5p-3, -0x1.3ec3bf7f09d08p-4,
        0x1.01fffa65118c8p-1, 0x1.ddf852bf70c00p-2, 0x1.9851aecd72ae5p-1, -0x1.482d78f5794c5p-2, -0x1.49f682e4a020bp-5, 0x1.3772225a156dap-3, -0x1.4190319f58064p-4,
        0x1.040019c0b0556p-1, 0x1.e127dfa2ba200p-2, 0x1.9709308c17a55p-1, -0x1.48a59957a7efdp-2, -0x1.369762648f2bbp-5, 0x1.344ab592569b1p-3, -0x1.4431803752ddbp-4,
        0x1.05fffc24501dbp-1, 0x1.e4547a495bcc0p-2, 0x1.95c064f225b79p-1, -0x1.491672163f5b8p-2, -0x1.236d34b79b89fp-5, 0x1.311d4b530b7bep-3, -0x1.46a844d931476p-4,
        0x1.07ffe865125fcp-1, 0x1.e77e92a5fad60p-2, 0x1.947725c13b0eap-1, -0x1.498026f33abcap-2, -0x1.1075ade947c6bp-5, 0x1.2de9dd8d5e01bp-3, -0x1.48f51ca17ca60p-4,
        0x1.0a002107eac25p-1, 0x1.eaa6908243180p-2, 0x1.932d4f339824bp-1, -0x1.49e2d7145f475p-2, -0x1.fb5d800571424p-6, 0x1.2ab0685d1cf84p-3, -0x1.4b18a7dbbbabep-4,
        0x1.0bfff7376e5d4p-1, 0x1.edcb5f79ff560p-2, 0x1.91e358ee1b492p-1, -0x1.4a3e749498453p-2, -0x1.d63e4be685c6fp-6, 0x1.27726c4b1f032p-3, -0x1.4d1389e6ecc3ap-4,
        0x1.0dffe1715ee2ep-1, 0x1.f0edb9be1bb80p-2, 0x1.9098fd993bd60p-1, -0x1.4a9329b84e907p-2, -0x1.b185ae07dba5ep-6, 0x1.242f8f2d7a804p-3, -0x1.4ee668ddaa340p-4,
        0x1.100017f3d776cp-1, 0x1.f40df6119e100p-2, 0x1.8f4e1fb44bcfbp-1, -0x1.4ae1116e3467ep-2, -0x1.8d304cf368422p-6, 0x1.20e7d736708aep-3, -0x1.5091ed7b3658dp-4,
        0x1.11ffefd8c7b65p-1, 0x1.f72b08fd21560p-2, 0x1.8e0334770fb0ap-1, -0x1.4b2825c0f6783p-2, -0x1.694ac7ffe0364p-6, 0x1.1d9cbe529bf4cp-3, -0x1.5216c2c73e5f0p-4,
        0x1.14000afa3ee71p-1, 0x1.fa45ee3324d60p-2, 0x1.8cb7d9ff684dfp-1, -0x1.4b68917add34dp-2, -0x1.45ca367276e70p-6, 0x1.1a4d9a1fbf3b1p-3, -0x1.537595fba2374p-4,
        0x1.15fff73336187p-1, 0x1.fd5df3de48d00p-2, 0x1.8b6c60cbe3546p-1, -0x1.4ba259b291bcbp-2, -0x1.22b6f5fb712ccp-6, 0x1.16fb855e28b0bp-3, -0x1.54af1633f423cp-4,
        0x1.17fff6c447b82p-1, 0x1.0039c0208ecc0p-1, 0x1.8a20a48f15926p-1, -0x1.4bd59a5808ac3p-2, -0x1.000cd5eef6f2ap-6, 0x1.13a66ebe54aa7p-3, -0x1.55c3f45420ce4p-4,
        0x1.19fffae932b61p-1, 0x1.01c33e0091bc0p-1, 0x1.88d4b55664e00p-1, -0x1.4c026579f5abbp-2, -0x1.bb9a68797c32ap-7, 0x1.104ec95d4f64ep-3, -0x1.56b4e2bbc325ep-4,
        0x1.1bfffba12ae50p-1, 0x1.034b6d3aba020p-1, 0x1.87889ebdccf04p-1, -0x1.4c28ce6d463c1p-2, -0x1.77f1cb36211fcp-7, 0x1.0cf4fb90b11e7p-3, -0x1.5782952dcbe1ap-4,
        0x1.1e0014b459e41p-1, 0x1.04d262dc05800p-1, 0x1.863c551625b6ap-1, -0x1.4c48eaffdd399p-2, -0x1.351cb603059cap-7, 0x1.09992de65d0d9p-3, -0x1.582dc087bb367p-4,
        0x1.2000032306f33p-1, 0x1.0657ebafb6ce0p-1, 0x1.84f00a1e2eec3p-1, -0x1.4c62cb79ec8c6p-2, -0x1.e6488d95de8d1p-8, 0x1.063c2661df241p-3, -0x1.58b71aaa63badp-4,
        0x1.22000d30a486cp-1, 0x1.07dc3d2165080p-1, 0x1.83a3966b3e5bfp-1, -0x1.4c7687de04deep-2, -0x1.63ff7800f052fp-8, 0x1.02ddc28f35eddp-3, -0x1.591f5a351cf91p-4,
        0x1.23ffe215e03fcp-1, 0x1.095f19f380a00p-1, 0x1.8257348be5f3fp-1, -0x1.4c8431b793f77p-2, -0x1.c6e63625993b8p-9, 0x1.fefdb8c5e4b3bp-4, -0x1.5967366fe9ca7p-4,
        0x1.260006833d65dp-1, 0x1.0ae0e6496a8c0p-1, 0x1.810a945b44aa3p-1, -0x1.4c8be055b407ap-2, -0x1.920a7ae83f0a4p-10, 0x1.f83dc860a6a5ep-4, -0x1.598f670d98ee7p-4,
        0x1.28000e82d4d50p-1, 0x1.0c615095f5300p-1, 0x1.7fbe01e9337b7p-1, -0x1.4c8da573c6f6ap-2, 0x1.8b6c7c50f565dp-12, 0x1.f17dbc9c4b6cap-4, -0x1.5998a45d6dae0p-4,
        0x1.29fff203b6a0bp-1, 0x1.0de0530852720p-1, 0x1.7e7188520538dp-1, -0x1.4c899668c6963p-2, 0x1.286ecbeca8ab0p-9, 0x1.eabe49b6ac5bdp-4, -0x1.5983a575a9684p-4,
        0x1.2c001e91a9d93p-1, 0x1.0f5e3f7817a20p-1, 0x1.7d24e63a45d97p-1, -0x1.4c7fc5f83c46dp-2, 0x1.0e1995d9c800ap-8, 0x1.e3fe93721a8e0p-4, -0x1.59512377da840p-4,
        0x1.2dfffc6fb4948p-1, 0x1.10daa4ce36040p-1, 0x1.7bd883e39011fp-1, -0x1.4c704b5eae11fp-2, 0x1.86398192c622bp-8, 0x1.dd412b62ba357p-4, -0x1.5901d5f0e020ep-4,
        0x1.2ffff39cb4eedp-1, 0x1.1255d0970ad60p-1, 0x1.7a8c2365b7a9bp-1, -0x1.4c5b38925f532p-2, 0x1.fcb03785e3070p-8, 0x1.d68540eedf3b3p-4, -0x1.58967479c252ap-4,
        0x1.31ffe002e31cbp-1, 0x1.13cfa81fd3780p-1, 0x1.793fe1bbe9667p-1, -0x1.4c40a3046f4c7p-2, 0x1.38bae8f5e6bf1p-7, 0x1.cfcbd83775c98p-4, -0x1.580fb62e887abp-4,
        0x1.34000edc7bffdp-1, 0x1.1548644d05200p-1, 0x1.77f39244a1da5p-1, -0x1.4c2099fb764c1p-2, 0x1.724e2851b0be5p-7, 0x1.c9147507c76e0p-4, -0x1.576e519c7f0abp-4,
        0x1.36001ce042830p-1, 0x1.16bfbc1656ae0p-1, 0x1.76a77ad3b2b77p-1, -0x1.4bfb374aac296p-2, 0x1.ab07005b229c2p-7, 0x1.c260e87dca54bp-4, -0x1.56b2fc90df763p-4,
        0x1.37ffe89b8fc54p-1, 0x1.1835977d0ba80p-1, 0x1.755bb660caa3dp-1, -0x1.4bd09308bb975p-2, 0x1.e2e26fe0a1240p-7, 0x1.bbb2218790f26p-4, -0x1.55de6c094f3dap-4,
        0x1.3a0019b4da842p-1, 0x1.19aa7100cd140p-1, 0x1.740fdd801f889p-1, -0x1.4ba0b2c32c656p-2, 0x1.0cf998eca44a2p-6, 0x1.b5066c9863443p-4, -0x1.54f15406672b5p-4,
        0x1.3c000ce6b63e8p-1, 0x1.1b1dd1d0b0ae0p-1, 0x1.72c45f28670e6p-1, -0x1.4b6bb92422e2ep-2, 0x1.28141a0d32146p-6, 0x1.ae60637452321p-4, -0x1.53ec677d91f56p-4,
        0x1.3dfff114a2607p-1, 0x1.1c8fdc6ff6f20p-1, 0x1.71792206847a7p-1, -0x1.4b31b669bd306p-2, 0x1.42c3a04ffd28ap-6, 0x1.a7bfde7fc0825p-4, -0x1.52d0582f471bap-4,
        0x1.3ffffc1da9b7dp-1, 0x1.1e00b7f2e8840p-1, 0x1.702e084371133p-1, -0x1.4af2b8012fbe4p-2, 0x1.5d0b4bfc47f4bp-6, 0x1.a1249d80ab6c5p-4, -0x1.519dd69a4108dp-4,
        0x1.41ffee11d9c33p-1, 0x1.1f70367c3ec20p-1, 0x1.6ee34026a76a0p-1, -0x1.4aaed96514b12p-2, 0x1.76e8307ba2905p-6, 0x1.9a8fe261a1221p-4, -0x1.505591d552ba0p-4,
        0x1.43ffffa174676p-1, 0x1.20de80faff860p-1, 0x1.6d98a9ea6d162p-1, -0x1.4a6626b927b3bp-2, 0x1.905d8f84adbb0p-6, 0x1.94015dd484db5p-4, -0x1.4ef83783eef44p-4,
        0x1.45fff0d457fa4p-1, 0x1.224b69f675300p-1, 0x1.6c4e73a093351p-1, -0x1.4a18bcbf2bff8p-2, 0x1.a968a84bb8c16p-6, 0x1.8d7a493fbb975p-4, -0x1.4d8673b37e4fbp-4,
        0x1.47ffe8f910e57p-1, 0x1.23b70dd92b840p-1, 0x1.6b04889b04359p-1, -0x1.49c6a974b07ffp-2, 0x1.c20be25f20251p-6, 0x1.86fa882e9673dp-4, -0x1.4c00f0d12f550p-4,
        0x1.4a0017323fc6bp-1, 0x1.25218e34e3420p-1, 0x1.69bacf277fe27p-1, -0x1.496ff7f856abap-2, 0x1.da49e9928150cp-6, 0x1.8081e3eb66a26p-4, -0x1.4a68578ab06c5p-4,
        0x1.4c000b1bf0500p-1, 0x1.268a9bd8b2c80p-1, 0x1.6871942abbd42p-1, -0x1.4914cec74e64ap-2, 0x1.f21ded0c3eeecp-6, 0x1.7a1225b30aa05p-4, -0x1.48bd4ec53ef43p-4,
        0x1.4e0011d07207bp-1, 0x1.27f26da64f7a0p-1, 0x1.6728aa7cfbeb2p-1, -0x1.48b533fcbb247p-2, 0x1.04c60a7354a41p-5, 0x1.73aaaeff6f27ap-4, -0x1.47007b81a6bb2p-4,
        0x1.4fffe5f36eb46p-1, 0x1.2958d35ddd180p-1, 0x1.65e04307b6af3p-1, -0x1.48514828bb6e6p-2, 0x1.1048e48993ed9p-5, 0x1.6d4cb468d7c59p-4, -0x1.453280d484989p-4,
        0x1.520012afdf759p-1, 0x1.2abe2eb1c3280p-1, 0x1.649808dc5daadp-1, -0x1.47e902c11e3b7p-2, 0x1.1b9ae88e1b343p-5, 0x1.66f6cff4501bfp-4, -0x1.4353ffcd6b8dep-4,
        0x1.54001dfdb2423p-1, 0x1.2c222ab0402c0p-1, 0x1.63504e7e657fbp-1, -0x1.477c8eee53fa9p-2, 0x1.26b9a696cd845p-5, 0x1.60aad6a3aa6efp-4, -0x1.416597704e1f4p-4,
        0x1.55ffe72d2a74fp-1, 0x1.2d84b16be7240p-1, 0x1.62092ce54aedep-1, -0x1.470c07b764156p-2, 0x1.31a4c4d9abee7p-5, 0x1.5a697a899a63dp-4, -0x1.3f67e49fa7fb1p-4,
        0x1.58000ee716c33p-1, 0x1.2ee63284f3fe0p-1, 0x1.60c24181c5720p-1, -0x1.46975c383b0c1p-2, 0x1.3c5ffc40a1a5ap-5, 0x1.543110b7b3b72p-4, -0x1.3d5b821700401p-4,
        0x1.59fff9825cd2ap-1, 0x1.304642defcf40p-1, 0x1.5f7bf3c14a317p-1, -0x1.461ec227a4cdep-2, 0x1.46e856da8d837p-5, 0x1.4e03c6162f4c8p-4, -0x1.3b410857f5976p-4,
        0x1.5bffdfe2a42cdp-1, 0x1.31a50a5110dc0p-1, 0x1.5e36233cf1268p-1, -0x1.45a23f68b7dbcp-2, 0x1.513f5de40f0e9p-5, 0x1.47e12de05901ep-4, -0x1.39190da5cabb5p-4,
        0x1.5e00057330799p-1, 0x1.3302b75253480p-1, 0x1.5cf0a901da45ap-1, -0x1.4521d552754cfp-2, 0x1.5b66bbbf000bbp-5, 0x1.41c8bd2baf7b2p-4, -0x1.36e425f53241ap-4,
        0x1.600014d6055dap-1, 0x1.345f0ff2eda60p-1, 0x1.5babbf2ea5900p-1, -0x1.449dab2008754p-2, 0x1.655d118f56fbbp-5, 0x1.3bbbb89a0c1b2p-4, -0x1.34a2e2e8d60fcp-4,
        0x1.620012c3809cbp-1, 0x1.35ba1812d5040p-1, 0x1.5a676671e49e9p-1, -0x1.4415d230e6216p-2, 0x1.6f22d6b05c7f7p-5, 0x1.35ba4cfe6b72bp-4, -0x1.3255d3c3bfa3bp-4,
        0x1.6400087b47eccp-1, 0x1.3713d69715580p-1, 0x1.59239c8fb0e69p-1, -0x1.438a5a5bd1f6ep-2, 0x1.78b897f9b13cfp-5, 0x1.2fc4974f57c8fp-4, -0x1.2ffd8566caacap-4,
        0x1.66000a746397fp-1, 0x1.386c59d968940p-1, 0x1.57e0583073c58p-1, -0x1.42fb4fe3d0083p-2, 0x1.821f14b9e1eebp-5, 0x1.29da91952ee82p-4, -0x1.2d9a8245866a8p-4,
        0x1.68000e4e3094bp-1, 0x1.39c39b5fe3900p-1, 0x1.569da36dd131ep-1, -0x1.4268c74778fe0p-2, 0x1.8b5679ab0310fp-5, 0x1.23fc8f2e43205p-4, -0x1.2b2d526483573p-4,
        0x1.6a001e2e37787p-1, 0x1.3b19a27d52620p-1, 0x1.555b7b5d865cdp-1, -0x1.41d2cf1600cd3p-2, 0x1.945f54b79e859p-5, 0x1.1e2aa46a0b02dp-4, -0x1.28b67b508a35bp-4,
        0x1.6bffe0df4bbfbp-1, 0x1.3c6e346f2b6e0p-1, 0x1.541a1b658afbep-1, -0x1.41399388da137p-2, 0x1.9d387e5b3c2bap-5, 0x1.18660173397f9p-4, -0x1.2636801db4945p-4,
        0x1.6dfffea406ceap-1, 0x1.3dc1c1bb3d400p-1, 0x1.52d91d33ffe8ep-1, -0x1.409cf36bcffe9p-2, 0x1.a5e54174405afp-5, 0x1.12acedc041806p-4, -0x1.23ade160d6557p-4,
        0x1.70000ed01ea65p-1, 0x1.3f14054e51400p-1, 0x1.5198c5c8b9119p-1, -0x1.3ffd1f2ea4ff7p-2, 0x1.ae643308c81cdp-5, 0x1.0d00c1960aaf7p-4, -0x1.211d1d2f50d25p-4,
        0x1.7200200d515ebp-1, 0x1.40650983bb3e0p-1, 0x1.50590f2175c71p-1, -0x1.3f5a2361bb15cp-2, 0x1.b6b5f9b536afcp-5, 0x1.07617a731624dp-4, -0x1.1e84af1a8c054p-4,
        0x1.740011323de6dp-1, 0x1.41b4b9483e720p-1, 0x1.4f1a11027ba01p-1, -0x1.3eb41bb978c8fp-2, 0x1.beda77765626ap-5, 0x1.01cf997f58c8ap-4, -0x1.1be5103074348p-4,
        0x1.75fff25cab4cap-1, 0x1.430320001d5c0p-1, 0x1.4ddbc4573fb6cp-1, -0x1.3e0b141f21d2ap-2, 0x1.c6d25d1bda00fp-5, 0x1.f89625935ee68p-5, -0x1.193eb6f8e0689p-4,
        0x1.77ffe90921f76p-1, 0x1.445056cc6af00p-1, 0x1.4c9e14cffbdaep-1, -0x1.3d5f10b247ec4p-2, 0x1.ce9ea943f4516p-5, 0x1.eda73f24a8af1p-5, -0x1.16921776aac42p-4,
        0x1.79ffe47b2f83bp-1, 0x1.459c535c19f20p-1, 0x1.4b610fc8f20bdp-1, -0x1.3cb0273df2a0dp-2, 0x1.d63f823c5d6dep-5, 0x1.e2d319c5116abp-5, -0x1.13dfa326e2972p-4,
        0x1.7bfff2f1e79a9p-1, 0x1.46e71f84df5c0p-1, 0x1.4a24af586b1bdp-1, -0x1.3bfe62ef81e5bp-2, 0x1.ddb58738896f0p-5, 0x1.d819a2515de78p-5, -0x1.1127c9026fdd0p-4,
        0x1.7e001973c8d05p-1, 0x1.4830bf0fb9580p-1, 0x1.48e8f3466b08ep-1, -0x1.3b49d1c53a01ap-2, 0x1.e501325103eedp-5, 0x1.cd7af5290f4afp-5, -0x1.0e6af57ef003bp-4,
        0x1.7ffff69efc092p-1, 0x1.4978f431c3800p-1, 0x1.47ae1a3e1064ap-1, -0x1.3a92a666c50c4p-2, 0x1.ec2194098a4bep-5, 0x1.c2f942eee57e0p-5, -0x1.0ba99290d5730p-4,
        0x1.82001c52b5232p-1, 0x1.4ac01d2b83340p-1, 0x1.4673cd31b7cf5p-1, -0x1.39d8bc67d05f0p-2, 0x1.f31922a81b5d5p-5, 0x1.b891b8aa20e90p-5, -0x1.08e407adcefd6p-4,
        0x1.84000bd4d4e3fp-1, 0x1.4c05e9b1dbc60p-1, 0x1.453a5c8d629f7p-1, -0x1.391c513e9ef47p-2, 0x1.f9e6917383d6bp-5, 0x1.ae471278e21b9p-5, -0x1.061ab9cf54d10p-4,
        0x1.860018c869cbdp-1, 0x1.4d4a8fd2285a0p-1, 0x1.4401979b82471p-1, -0x1.385d55c3e2929p-2, 0x1.0045b7b2c8ff2p-4, 0x1.a417c39d7ca4fp-5, -0x1.034e0b767b7d4p-4,
        0x1.87ffeb5db3710p-1, 0x1.4e8dd8b93bca0p-1, 0x1.42c9b66c6e6bfp-1, -0x1.379bfa32ee2a1p-2, 0x1.038386187fe0fp-4, 0x1.9a05a8b3a0b33p-5, -0x1.007e5caee03a9p-4,
        0x1.8a000863c77e3p-1, 0x1.4fd018fcd1e80p-1, 0x1.41926a8a8093fp-1, -0x1.36d81b5ee344dp-2, 0x1.06adc2841f292p-4, 0x1.900e42484560bp-5, -0x1.fb58162792f0ap-5,
        0x1.8bfff0ed982afp-1, 0x1.5111016e28ac0p-1, 0x1.405c0389112eep-1, -0x1.3611f89d38dc7p-2, 0x1.09c3db450b9f7p-4, 0x1.86342312d0c4ap-5, -0x1.f5aee3a6ca012p-5,
        0x1.8e00002c3aeaep-1, 0x1.5250cc0ab0a40p-1, 0x1.3f264c65593c5p-1, -0x1.35497d82be900p-2, 0x1.0cc6968546d39p-4, 0x1.7c759db8499fdp-5, -0x1.f001d36a32337p-5,
        0x1.90000ecbfa97bp-1, 0x1.538f60e8d4ee0p-1, 0x1.3df15f4119333p-1, -0x1.347ec7d2149f4p-2, 0x1.0fb5efa921d3cp-4, 0x1.72d3869693e89p-5, -0x1.ea51923a0f5f3p-5,
        0x1.91fffd251c01cp-1, 0x1.54ccad3f3bd20p-1, 0x1.3cbd51554dd15p-1, -0x1.33b1f2bc94245p-2, 0x1.1291f2fc4c3f6p-4, 0x1.694e81b7a765cp-5, -0x1.e49ec826e86f6p-5,
        0x1.94001d90af4e6p-1, 0x1.5608e4d4ec640p-1, 0x1.3b89f3445ef72p-1, -0x1.32e2eb7bbd79ap-2, 0x1.155b4e401d071p-4, 0x1.5fe513a256f1cp-5, -0x1.deea1890ff662p-5,
        0x1.9600104fd6c17p-1, 0x1.5743cd5673c20p-1, 0x1.3a57809ebc6e2p-1, -0x1.3211e6da5039cp-2, 0x1.1811b4e62286bp-4, 0x1.5699071bece9dp-5, -0x1.d934223911641p-5,
        0x1.980002d214b82p-1, 0x1.587d83b0d6120p-1, 0x1.3925e01eaac3ep-1, -0x1.313ee08425504p-2, 0x1.1ab5a02bdb571p-4, 0x1.4d6989ebd70b8p-5, -0x1.d37d7f482965ap-5,
        0x1.99ffdeb980651p-1, 0x1.59b5fb16ba7a0p-1, 0x1.37f5210b1ab7ap-1, -0x1.3069ef993d676p-2, 0x1.1d472cded25a8p-4, 0x1.445702d0abd9ap-5, -0x1.cdc6c56221aa1p-5,
        0x1.9bfffe5504053p-1, 0x1.5aed6b55de6a0p-1, 0x1.36c50fa91c51ep-1, -0x1.2f92fbe311e56p-2, 0x1.1fc705be3af05p-4, 0x1.3b5fdacd5cdc7p-5, -0x1.c81085adbb9b8p-5,
        0x1.9e0016e60a234p-1, 0x1.5c23a79acd480p-1, 0x1.3595da5fab2eap-1, -0x1.2eba31ddeceeap-2, 0x1.2235035736518p-4, 0x1.3285622f9fd28p-5, -0x1.c25b4ce8b2259p-5,
        0x1.9ffffb685741bp-1, 0x1.5d5895ad40460p-1, 0x1.34679d832b8d3p-1, -0x1.2ddfb230eda41p-2, 0x1.24912b23c0ba2p-4, 0x1.29c854c4e86dap-5, -0x1.bca7a37002a55p-5,
        0x1.a20019d59b943p-1, 0x1.5e8c78c187ea0p-1, 0x1.333a19ede2183p-1, -0x1.2d035b0043779p-2, 0x1.26dc37ab9110cp-4, 0x1.2126c959cfc0ep-5, -0x1.b6f60d556233ep-5,
        0x1.a3fffbe9e153fp-1, 0x1.5fbf0a9c08ae0p-1, 0x1.320d96f7861aap-1, -0x1.2c256c2200f18p-2, 0x1.2915da6795293p-4, 0x1.18a2b256a8fdep-5, -0x1.b1470a67a4e89p-5,
        0x1.a5ffe7a23a1cep-1, 0x1.60f0763200600p-1, 0x1.30e1ed13d395ep-1, -0x1.2b45d44403932p-2, 0x1.2b3e9c967f013p-4, 0x1.103ad35d002b8p-5, -0x1.ab9b16496a8f1p-5,
        0x1.a800157f250b8p-1, 0x1.6220ddd6453a0p-1, 0x1.2fb6fcfffcc1ep-1, -0x1.2a6486f8d8291p-2, 0x1.2d56f03654cc3p-4, 0x1.07ee34bb6e7a6p-5, -0x1.a5f2a87992f03p-5,
        0x1.aa000dd839d49p-1, 0x1.634ffb412c9a0p-1, 0x1.2e8d0e2d59e01p-1, -0x1.2981c5467cfddp-2, 0x1.2f5e8ff1fadb5p-4, 0x1.ff7d6a3ba803cp-6, -0x1.a04e346af8db7p-5,
        0x1.ac000770df220p-1, 0x1.647defef70020p-1, 0x1.2d640220aff7fp-1, -0x1.289d836f9e74fp-2, 0x1.3155ee509140ap-4, 0x1.ef56b61ab0b7fp-6, -0x1.9aae298ce391fp-5,
        0x1.ae001125bbe48p-1, 0x1.65aac57a24d20p-1, 0x1.2c3bd1bfb3559p-1, -0x1.27b7c6dde55ddp-2, 0x1.333d515c4c270p-4, 0x1.df67a9bac4ecfp-6, -0x1.9512f363a972bp-5,
        0x1.afffe7c321839p-1, 0x1.66d65569b83c0p-1, 0x1.2b14a53fbf8d9p-1, -0x1.26d0b9cfa03cep-2, 0x1.3514b2caa2e0cp-4, 0x1.cfb224597be9ap-6, -0x1.8f7cf99110022p-5,
        0x1.b1ffe75486924p-1, 0x1.6800d68cefb40p-1, 0x1.29ee48e6aa814p-1, -0x1.25e83e8afa7ebp-2, 0x1.36dc9fb0e8ac8p-4, 0x1.c0331ad5d66cap-6, -0x1.89ec9fedb1e8bp-5,
        0x1.b40015fb8deb8p-1, 0x1.692a4d137c500p-1, 0x1.28c8babff668ep-1, -0x1.24fe5d8e71e0ap-2, 0x1.389551297317ap-4, 0x1.b0ea31d844655p-6, -0x1.846246914067dp-5,
        0x1.b6000386c27b9p-1, 0x1.6a5278cdf6fc0p-1, 0x1.27a43c5758db8p-1, -0x1.2413559cadce0p-2, 0x1.3a3e9ee34ae91p-4, 0x1.a1da81c5fff05p-6, -0x1.7ede49ec8aac6p-5,
        0x1.b8000d1efddb3p-1, 0x1.6b7990accb660p-1, 0x1.268099983aab2p-1, -0x1.2327076047e08p-2, 0x1.3bd90f132139bp-4, 0x1.9301058deb3e1p-6, -0x1.796102d194ce9p-5,
        0x1.b9ffe42cc4047p-1, 0x1.6c9f686445e60p-1, 0x1.255e0069f871fp-1, -0x1.2239a25461639p-2, 0x1.3d649a926c127p-4, 0x1.845fbc5a21f70p-6, -0x1.73eac68e20be6p-5,
        0x1.bc001951aeaadp-1, 0x1.6dc453c4e45a0p-1, 0x1.243c1ff6573b0p-1, -0x1.214aee38fa7e7p-2, 0x1.3ee1e5ea1330fp-4, 0x1.75f242bcce6dfp-6, -0x1.6e7be6f3902c5p-5,
        0x1.bdffe6616fe11p-1, 0x1.6ee7e27106fe0p-1, 0x1.231b697b587f0p-1, -0x1.205b5240fef32p-2, 0x1.4050944eb818cp-4, 0x1.67bde108160f9p-6, -0x1.6914b271d18adp-5,
        0x1.bffff54511c72p-1, 0x1.700a7643bbb40p-1, 0x1.21fb7e1823c8bp-1, -0x1.1f6a89a854f7ap-2, 0x1.41b1571f04837p-4, 0x1.59bd8bbd10f7cp-6, -0x1.63b5741f03711p-5,
        0x1.c2000c537593ep-1, 0x1.712bef36d6400p-1, 0x1.20dc7f754b2d5p-1, -0x1.1e78b9d24dbedp-2, 0x1.4304394f485e0p-4, 0x1.4bf29122a6884p-6, -0x1.5e5e73d2aa4e9p-5,
        0x1.c4000ddd35719p-1, 0x1.724c3d7fa3000p-1, 0x1.1fbe7f2a8b1bfp-1, -0x1.1d85fb25dddf6p-2, 0x1.44495d2e3b20fp-4, 0x1.3e5d67fcc1b30p-6, -0x1.590ff62d0d00fp-5,
        0x1.c6000402375b6p-1, 0x1.736b67dff3720p-1, 0x1.1ea1786c92387p-1, -0x1.1c92531ddfc58p-2, 0x1.4580ff8b6cbc2p-4, 0x1.30fd700ce998ep-6, -0x1.53ca3cb299e5fp-5,
        0x1.c7fff19904fe4p-1, 0x1.748970f395860p-1, 0x1.1d856a825ba33p-1, -0x1.1b9dca75e0fc5p-2, 0x1.46ab579f8fd7dp-4, 0x1.23d23a5a90afep-6, -0x1.4e8d85d2f574bp-5,
        0x1.c9ffef9e2409dp-1, 0x1.75a6679e7f1c0p-1, 0x1.1c6a48740d2e9p-1, -0x1.1aa85f198392cp-2, 0x1.47c8a808c583ap-4, 0x1.16dac857f2526p-6, -0x1.495a0d0477576p-5,
        0x1.cc001e038ef72p-1, 0x1.76c25e6815140p-1, 0x1.1b50019bdadf8p-1, -0x1.19b20b4a469aep-2, 0x1.48d9342387ea2p-4, 0x1.0a15f7305baf5p-6, -0x1.44300acae4e17p-5,
        0x1.cdffeeb72037fp-1, 0x1.77dd07a7a4aa0p-1, 0x1.1a36e4f1f6702p-1, -0x1.18bb1d0992cf8p-2, 0x1.49dce5aa4990dp-4, 0x1.fb0dd63759665p-7, -0x1.3f0fb4d2f0c0fp-5,
        0x1.cffffea4839edp-1, 0x1.78f6bb17088c0p-1, 0x1.191e9cf32122fp-1, -0x1.17c35220400acp-2, 0x1.4ad440a159641p-4, 0x1.e252c80894ca9p-7, -0x1.39f93df89c265p-5,
        0x1.d1ffdec3ec8b2p-1, 0x1.7a0f3c8c6c880p-1, 0x1.18076729f01d6p-1, -0x1.16cae98515540p-2, 0x1.4bbf41b0933ffp-4, 0x1.c9ff5e09a60cdp-7, -0x1.34ecd662a5704p-5,
        0x1.d3fff7084edd4p-1, 0x1.7b26c5f02f220p-1, 0x1.16f10b9973206p-1, -0x1.15d1b9e1e0a54p-2, 0x1.4c9e4ac2c9a30p-4, 0x1.b20ddefce76ccp-7, -0x1.2feaab888bc37p-5,
        0x1.d5ffe8d728e7cp-1, 0x1.7c3d2488d7e80p-1, 0x1.15dbbe622a5a7p-1, -0x1.14d7fa305ceb2p-2, 0x1.4d716417bf1c7p-4, 0x1.9a81ee19fe239p-7, -0x1.2af2e84ddad07p-5,
        0x1.d7fff70aa3b03p-1, 0x1.7d527db239580p-1, 0x1.14c75be4fea01p-1, -0x1.13dd92ad706aap-2, 0x1.4e38db49d32aap-4, 0x1.8357a37df2b6dp-7, -0x1.2605b507cd77bp-5,
        0x1.d9fff1434fba3p-1, 0x1.7e66b82c8a720p-1, 0x1.13b3fed9b7fedp-1, -0x1.12e2a3ac9d646p-2, 0x1.4ef4ce7b01cf5p-4, 0x1.6c905d25fd52dp-7, -0x1.21233798666efp-5,
        0x1.dbffea8c8de8cp-1, 0x1.7f79df4a0a520p-1, 0x1.12a19d7fc2119p-1, -0x1.11e72c6be19dfp-2, 0x1.4fa57634e1b91p-4, 0x1.562a647f96df5p-7, -0x1.1c4b9373af599p-5,
        0x1.de00026573df5p-1, 0x1.808c04dbcb960p-1, 0x1.119027903e4b9p-1, -0x1.10eb25cdfed06p-2, 0x1.504b0cca681fap-4, 0x1.402386f3cde09p-7, -0x1.177ee9ba8fa6ap-5,
        0x1.dfffe35009b66p-1, 0x1.819cfc2cb5340p-1, 0x1.107fcb1c942b5p-1, -0x1.0feec230d7d92p-2, 0x1.50e5a75c5b4f1p-4, 0x1.2a7e8e3c139d8p-7, -0x1.12bd593fa642bp-5,
        0x1.e2000492d4c68p-1, 0x1.82ad05ccb8680p-1, 0x1.0f704928e55dfp-1, -0x1.0ef1cee0b0721p-2, 0x1.51759937bfb74p-4, 0x1.153592bc9fddbp-7, -0x1.0e06fea1d1824p-5,
        0x1.e40009412bb65p-1, 0x1.83bbf14001a60p-1, 0x1.0e61d37f485dap-1, -0x1.0df481b2bd37dp-2, 0x1.51faf64024d14p-4, 0x1.004b99b849698p-7, -0x1.095bf450a2434p-5,
        0x1.e5fff4758ef2fp-1, 0x1.84c9c1531c180p-1, 0x1.0d5468b7fece7p-1, -0x1.0cf6e105bfe1ep-2, 0x1.5275ef9c5e03ap-4, 0x1.d77f217aa1137p-8, -0x1.04bc52a6891e1p-5,
        0x1.e8000380f819fp-1, 0x1.85d6974ccc060p-1, 0x1.0c47e8f1da5b5p-1, -0x1.0bf8d62ad700fp-2, 0x1.52e6c1f3fbc2bp-4, 0x1.af1c3ee24ad7dp-8, -0x1.00282fece26c9p-5,
        0x1.ea000a6d8cb7bp-1, 0x1.86e25d00e3a60p-1, 0x1.0b3c6ba314d62p-1, -0x1.0afa7e7cb2d84p-2, 0x1.534d908e9071fp-4, 0x1.877044ce5e5c9p-8, -0x1.f73f40eb7c9d5p-6,
        0x1.ec0005a13ba60p-1, 0x1.87ed119b163e0p-1, 0x1.0a31f2ebb7ad7p-1, -0x1.09fbe33a3fce1p-2, 0x1.53aa889d9af5dp-4, 0x1.60799f7f7040bp-8, -0x1.ee456d3f0b3fbp-6,
        0x1.edfff58f8dd18p-1, 0x1.88f6b6681ca80p-1, 0x1.09287ec4360b3p-1, -0x1.08fd0b7ce07e5p-2, 0x1.53fdd7bdedd3fp-4, 0x1.3a36670c52e66p-8, -0x1.e56305dca7315p-6,
        0x1.effffbe033400p-1, 0x1.89ff5dd4d7960p-1, 0x1.081ffdffe15bdp-1, -0x1.07fdedae56c0fp-2, 0x1.5447af84d6f5dp-4, 0x1.14a247982941ep-8, -0x1.dc98281e68835p-6,
        0x1.f2001e6b5125dp-1, 0x1.8b070bbe88160p-1, 0x1.07186df7122e2p-1, -0x1.06fe8de905325p-2, 0x1.54883b5deec7ap-4, 0x1.df762b4a186d5p-9, -0x1.d3e4ede20f495p-6,
        0x1.f3ffdf770e0dbp-1, 0x1.8c0d809e96380p-1, 0x1.06120f5a576a9p-1, -0x1.05ff31d2912ffp-2, 0x1.54bf98cd1001fp-4, 0x1.970fc6e90dc16p-9, -0x1.cb496d8eb587ep-6,
        0x1.f5ffe4e16da33p-1, 0x1.8d13129bccdc0p-1, 0x1.050c8d33ba4e9p-1, -0x1.04ff8d74c83d2p-2, 0x1.54ee0592bb252p-4, 0x1.4ff617193eeb5p-9, -0x1.c2c5ba459ac86p-6,
        0x1.f80004576ff2ep-1, 0x1.8e17acce443a0p-1, 0x1.0407fd8a97b6cp-1, -0x1.03ffbc91b3e55p-2, 0x1.5513a5f3357f7p-4, 0x1.0a2ba14c92b53p-9, -0x1.ba59e3e70df71p-6,
        0x1.f9fff39b6a330p-1, 0x1.8f1b2a7f515a0p-1, 0x1.0304863064158p-1, -0x1.02ffeacbaada8p-2, 0x1.55309f27448c0p-4, 0x1.8b6d64850006bp-10, -0x1.b205f742323dfp-6,
        0x1.fc001aa76c0b9p-1, 0x1.901dc15d66d80p-1, 0x1.0201f28d9b4aap-1, -0x1.01ffea98d4c38p-2, 0x1.55452089780f8p-4, 0x1.050b57f35c5bbp-10, -0x1.a9c9fe19247afp-6,
        0x1.fdffe39a592cap-1, 0x1.911f26d88a780p-1, 0x1.01008e40c6538p-1, -0x1.01000d31688dep-2, 0x1.55514e32f1816p-4, 0x1.02a154e1628d2p-11, -0x1.a1a5ff4faf5a0p-6,
        0x1.ff8018e92d1b0p-1, 0x1.91dfb9bb4bf00p-1, 0x1.003ffb884c5a9p-1, -0x1.003ff3876a954p-2, 0x1.555515539ddfbp-4, 0x1.007e77b95e6c2p-13, -0x1.9b9a718a3ba58p-6,
    };
    return tab;
}

namespace detail {
constexpr double kHpi = 0x1.921fb54442d18p+0, kHpi1 = 0x1.1a62633145c07p-54, kOpi = 0x1.921fb54442d18p+1, kOpi1 = 0x1.1a62633145c07p-53,
                 kQpi = 0x1.921fb54442d18p-1, kTqpi = 0x1.2d97c7f3321d2p+1;
// d3 + v (d5 + v (d7 + v (d9 + v (d11 + v d13))))   (e_atan2.c / s_atan.c, |u| < 1/16)
RDR_FN double atan_poly_small(double v) {
    double p = fma(v, 0x1.375f08b31cbcep-4, -0x1.7458022b13c25p-4);
    p = fma(v, p, 0x1.c71c6e5129a3bp-4);
    p = fma(v, p, -0x1.24924923f7603p-3);
    p = fma(v, p, 0x1.99999999997fdp-3);
    return fma(v, p, -0x1.5555555555555p-2);
}
RDR_FN const double *atan_row(double u) {
    const double two52 = 0x1p+52;
    int i = (int)(fma(u, 256.0, two52) - two52);
    return atan_table() + 7 * (i - 16);
}
} // namespace detail

// e_atan2.c: __ieee754_atan2
RDR_FN double atan2(double y, double x) {
    using namespace detail;
    const uint32_t ux = hi_word(x), dx = lo_word(x), uy = hi_word(y), dy = lo_word(y);
    if ((ux & 0x7ff00000u) == 0x7ff00000u && ((ux & 0x000fffffu) | dx) != 0) return x + y;
    if ((uy & 0x7ff00000u) == 0x7ff00000u && ((uy & 0x000fffffu) | dy) != 0) return y + y;
    if (uy == 0x00000000u) { if (dy == 0) return (ux & 0x80000000u) ? kOpi : 0.0; }
    else if (uy == 0x80000000u) { if (dy == 0) return (ux & 0x80000000u) ? -kOpi : -0.0; }
    if (x == 0) return (uy & 0x80000000u) ? -kHpi : kHpi;
    if (ux == 0x7ff00000u && dx == 0) {
        if (uy == 0x7ff00000u && dy == 0) return kQpi;
        if (uy == 0xfff00000u && dy == 0) return -kQpi;
        return (uy & 0x80000000u) ? -0.0 : 0.0;
    }
    if (ux == 0xfff00000u && dx == 0) {
        if (uy == 0x7ff00000u && dy == 0) return kTqpi;
        if (uy == 0xfff00000u && dy == 0) return -kTqpi;
        return (uy & 0x80000000u) ? -kOpi : kOpi;
    }
    if (uy == 0x7ff00000u && dy == 0) return kHpi;
    if (uy == 0xfff00000u && dy == 0) return -kHpi;

    double ax = (x < 0) ? -x : x, ay = (y < 0) ? -y : y;
    const int de = (int)(uy & 0x7ff00000u) - (int)(ux & 0x7ff00000u);
    if (de >= 59768832) return (y > 0) ? kHpi : -kHpi;
    if (de <= -59768832) {
        if (x > 0) return copysign(ay / ax, y);
        return (y > 0) ? kOpi : -kOpi;
    }
    const double twom500 = 0x1p-500, two500 = 0x1p+500;
    if (ax < twom500 || ay < twom500) { ax *= two500; ay *= two500; }
    if (ax > two500 || ay > two500) { ax *= twom500; ay *= twom500; }

    double u, du;
    if (ay < ax) {
        u = ay / ax;
        const double v = ax * u, vv = fma(ax, u, -v);
        du = ((ay - v) - vv) / ax;
    } else {
        u = ax / ay;
        const double v = ay * u, vv = fma(ay, u, -v);
        du = ((ax - v) - vv) / ay;
    }
    const double inv16 = 0.0625;
    double z;
    if (x > 0) {
        if (ay < ax) {                                  // (i) atan(ay / ax)
            if (u < inv16) {
                const double v = u * u;
                z = u + fma(u * v, atan_poly_small(v), du);
            } else {
                const double *c = atan_row(u);
                const double t3 = u - c[0];
                const double v = t3 + du;
                const double dv = (fabs(t3) > fabs(du)) ? ((t3 - v) + du) : ((du - v) + t3);
                const double q = fma(v, fma(v, fma(v, c[6], c[5]), c[4]), c[3]);
                const double zz = fma(v, c[2], fma(dv, c[2], (v * v) * q));
                z = c[1] + zz;
            }
        } else {                                        // (ii) pi/2 - atan(ax / ay)
            if (u < inv16) {
                const double v = u * u;
                const double zz = (u * v) * atan_poly_small(v);
                const double t2 = kHpi - u;
                const double cor = (fabs(kHpi) > fabs(u)) ? ((kHpi - t2) - u) : (kHpi - (u + t2));
                const double t3 = ((cor + kHpi1) - du) - zz;
                z = t2 + t3;
            } else {
                const double *c = atan_row(u);
                const double v = (u - c[0]) + du;
                const double p = fma(v, fma(v, fma(v, fma(v, c[6], c[5]), c[4]), c[3]), c[2]);
                const double zz = fma(-v, p, kHpi1);
                z = (kHpi - c[1]) + zz;
            }
        }
    } else if (ax < ay) {                               // (iii) pi/2 + atan(ax / ay)
        if (u < inv16) {
            const double v = u * u;
            const double zz = (v * u) * atan_poly_small(v);
            const double t2 = kHpi + u;
            const double cor = (fabs(kHpi) > fabs(u)) ? ((kHpi - t2) + u) : ((u - t2) + kHpi);
            const double t3 = ((cor + kHpi1) + du) + zz;
            z = t2 + t3;
        } else {
            const double *c = atan_row(u);
            const double v = (u - c[0]) + du;
            const double p = fma(v, fma(v, fma(v, fma(v, c[6], c[5]), c[4]), c[3]), c[2]);
            const double zz = fma(v, p, kHpi1);
            z = (kHpi + c[1]) + zz;
        }
    } else {                                            // (iv) pi - atan(ay / ax)
        if (u < inv16) {
            const double v = u * u;
            const double zz = (v * u) * atan_poly_small(v);
            const double t2 = kOpi - u;
            const double cor = (fabs(kOpi) > fabs(u)) ? ((kOpi - t2) - u) : (kOpi - (u + t2));
            const double t3 = ((cor + kOpi1) - du) - zz;
            z = t2 + t3;
        } else {
            const double *c = atan_row(u);
            const double v = (u - c[0]) + du;
            const double p = fma(v, fma(v, fma(v, fma(v, c[6], c[5]), c[4]), c[3]), c[2]);
            const double zz = fma(-v, p, kOpi1);
            z = (kOpi - c[1]) + zz;
        }
    }
    return copysign(z, y);
}

// s_atan.c: __atan
RDR_FN double atan(double x) {
    using namespace detail;
    const uint32_t ux = hi_word(x), dx = lo_word(x);
    if ((ux & 0x7ff00000u) == 0x7ff00000u && ((ux & 0x000fffffu) | dx) != 0) return x + x;
    const double u = (x < 0) ? -x : x;
    if (u < 1.0) {
        if (u < 0.0625) {
            if (u < 0x1.bb67ap-27) return x;
            const double v = x * x;
            return fma(x * v, atan_poly_small(v), x);
        }
        const double *c = atan_row(u);
        const double z = u - c[0];
        const double yy = fma(z, fma(z, fma(z, fma(z, c[6], c[5]), c[4]), c[3]), c[2]);
        return copysign(fma(z, yy, c[1]), x);
    }
    if (u < 16.0) {
        const double w = 1.0 / u;
        const double t1 = u * w, t2 = fma(u, w, -t1);
        const double *c = atan_row(w);
        const double z = fma((1.0 - t1) - t2, w, w - c[0]);
        double yy = fma(z, fma(z, fma(z, fma(z, c[6], c[5]), c[4]), c[3]), c[2]);
        yy = fma(-z, yy, kHpi1);
        return copysign((kHpi - c[1]) + yy, x);
    }
    if (u < 0x1.49ff2p+52) {
        const double w = 1.0 / u;
        const double v = w * w;
        const double t1 = u * w, t2 = fma(u, w, -t1);
        const double yy = (w * v) * atan_poly_small(v);
        const double ww = ((1.0 - t1) - t2) * w;
        const double t3 = kHpi - w;
        const double cor = (kHpi > fabs(w)) ? ((kHpi - t3) - w) : (kHpi - (w + t3));
        return copysign(t3 + (((cor + kHpi1) - ww) - yy), x);
    }
    return (x > 0) ? kHpi : -kHpi;
}

} // namespace gm
