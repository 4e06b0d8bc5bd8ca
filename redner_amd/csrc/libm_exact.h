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
// Range: every double (sin / cos of |x| >= 105414350 go through branred.c's reduction with 1800 bits of 2 / pi, which
// glibc builds without FMA: restated without).
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

// 2 / pi in 75 digits of 24 bits -- branred.h: toverp
RDR_FN const double *two_over_pi_table() {
    static const double tab[75] = {
        10680707.0, 7228996.0, 1387004.0, 2578385.0, 16069853.0, 12639074.0, 9804092.0, 4427841.0, 16666979.0, 11263675.0,
        12935607.0, 2387514.0, 4345298.0, 14681673.0, 3074569.0, 13734428.0, 16653803.0, 1880361.0, 10960616.0, 8533493.0,
        3062596.0, 8710556.0, 7349940.0, 6258241.0, 3772886.0, 3769171.0, 3798172.0, 8675211.0, 12450088.0, 3874808.0,
        9961438.0, 366607.0, 15675153.0, 9132554.0, 7151469.0, 3571407.0, 2607881.0, 12013382.0, 4155038.0, 6285869.0,
        7677882.0, 13102053.0, 15825725.0, 473591.0, 9065106.0, 15363067.0, 6271263.0, 9264392.0, 5636912.0, 4652155.0,
        7056368.0, 13614112.0, 10155062.0, 1944035.0, 9527646.0, 15080200.0, 6658437.0, 6231200.0, 6832269.0, 16767104.0,
        5075751.0, 3212806.0, 1398474.0, 7579849.0, 6349435.0, 12618859.0, 4703257.0, 12806093.0, 14477321.0, 2786137.0,
        12875403.0, 9837734.0, 14528324.0, 13719321.0, 343717.0,
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
// branred.c: one half of the argument times 2 / pi -- six 24-bit digits of 2 / pi from the position the half's exponent
// selects; integer parts beyond two bits dropped digit by digit.  -> fraction (b, bb) of a quarter turn, quarter count in sum
RDR_FN void branred_half(double xh, double &b, double &bb, double &sum) {
    const double tm24 = 0x1p-24, big = 0x1.8p+52, big1 = 0x1.8p+54;
    int k = (int)((hi_word(xh) >> 20) & 2047);
    k = (k - 450) / 24;
    if (k < 0) k = 0;
    double gor = bits2d((uint64_t)(0x63f00000u - (uint32_t)((k * 24) << 20)) << 32);       // 2^(576 - 24 k)
    const double *tv = two_over_pi_table() + k;
    double r[6];
    for (int i = 0; i < 6; ++i) { r[i] = xh * tv[i] * gor; gor *= tm24; }
    sum = 0;
    for (int i = 0; i < 3; ++i) { const double s = (r[i] + big) - big; sum += s; r[i] -= s; }
    double t = 0;
    for (int i = 0; i < 6; ++i) t += r[5 - i];
    bb = (((((r[0] - t) + r[1]) + r[2]) + r[3]) + r[4]) + r[5];
    double s = (t + big) - big;
    sum += s;
    t -= s;
    b = t + bb;
    bb = (t - b) + bb;
    s = (sum + big1) - big1;
    sum -= s;
}
// branred.c: __branred -- x = n pi/2 + (a + da) for 105414350 <= |x| < 2^1024
RDR_FN int branred(double x, double &a, double &da) {
    const double tm600 = 0x1p-600, split = 134217729.0, mp1 = 0x1.921fb58000000p+0, mp2 = -0x1.dde9740000000p-27;
    x *= tm600;
    double t = x * split;
    const double x1 = t - (t - x);
    const double x2 = x - x1;
    double b1, bb1, sum1, b2, bb2, sum2;
    branred_half(x1, b1, bb1, sum1);
    branred_half(x2, b2, bb2, sum2);
    double sum = sum1 + sum2;
    double b = b1 + b2;
    double bb = (fabs(b1) > fabs(b2)) ? (b1 - b) + b2 : (b2 - b) + b1;
    if (b > 0.5) { b -= 1.0; sum += 1.0; }
    else if (b < -0.5) { b += 1.0; sum -= 1.0; }
    double s = b + (bb + bb1 + bb2);
    t = ((b - s) + bb) + (bb1 + bb2);
    b = s * split;
    const double t1 = b - (b - s);
    const double t2 = s - t1;
    b = s * kHp0;
    bb = (((t1 * mp1 - b) + t1 * mp2) + t2 * mp1) + (t2 * mp2 + s * kHp1 + t * kHp0);
    s = b + bb;
    t = (b - s) + bb;
    a = s; da = t;
    return ((int)sum) & 3;
}
} // namespace detail

// s_sin.c: __sin
RDR_FN double sin(double x) {
    using namespace detail;
    const uint32_t k = hi_word(x) & 0x7fffffffu;
    if (k < 0x3e500000u) return x;                            // |x| < 2^-26
    if (k < 0x3feb6000u) return do_sin(x, 0.0);               // |x| < 0.855469
    if (k < 0x400368fdu) return copysign(do_cos(kHp0 - fabs(x), kHp1), x);   // |x| < 2.426265
    if (k < 0x419921fbu) { double a, da; int n = reduce_sincos(x, a, da); return do_sincos(a, da, n); }
    if (k < 0x7ff00000u) { double a, da; int n = branred(x, a, da); return do_sincos(a, da, n); }
    return x / x;                                             // inf, nan
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
    if (k < 0x7ff00000u) { double a, da; int n = branred(x, a, da); return do_sincos(a, da, n + 1); }
    return x / x;
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

// asin around 2568 / ~13 expansion points on [0.125, 0.96875): {point, series coefficients ..., asin(point) tail, head} -- glibc's
// asncs (asincos.tbl); first guesses of 1 / sqrt on [1, 2) -- inroot (root.tbl)
RDR_FN const double *asin_table() {
    static const double tab[2568] = {
        0x1.0400000000000p-3, 0x1.0216988994424p+0, 0x1.0a6a2b799b115p-4, 0x1.6ef15d57409a0p-3, 0x1.a141eaf52eaa0p-5, 0x1.75591abbbe261p-4, 0x1.72b51d206d88fp-5, 0x1.6b5955bb33e7dp-54,
        0x1.04b41a03e2700p-3, 0x1.0400000000000p+0, -0x1.e967766bbdc7cp-8, 0x1.0c00000000000p-3, 0x1.02386f9e23a56p+0, 0x1.1308c60fd0235p-4, 0x1.7099f14d16b02p-3, 0x1.afed627c01ee1p-5,
        0x1.79c6fdbcd5f98p-4, 0x1.8144a4084daacp-5, -0x1.7c09238d8505ep-55, 0x1.0cc5556c9f380p-3, 0x1.0400000000000p+0, -0x1.c79061dc5aa24p-8, 0x1.1400000000000p-3, 0x1.025b5b27141f6p+0,
        0x1.1bb1804ce7400p-4, 0x1.7251472907342p-3, 0x1.bec600bf4222cp-5, 0x1.7e61075b3736cp-4, 0x1.9024c5199c343p-5, -0x1.ae84c06b56f60p-55, 0x1.14d7a3defa070p-3, 0x1.0400000000000p+0,
        -0x1.a4a4d8ebe0a5cp-8, 0x1.1c00000000000p-3, 0x1.027f5c6de8f57p+0, 0x1.2464b345751e1p-4, 0x1.74178cf026805p-3, 0x1.cdcd840a9e0d6p-5, 0x1.83282eb1d9c38p-4, 0x1.9f590d7be707bp-5,
        -0x1.b976803a2a6d6p-53, 0x1.1ceb0e03b4870p-3, 0x1.0400000000000p+0, -0x1.80a392170a943p-8, 0x1.2400000000000p-3, 0x1.02a474c759796p+0, 0x1.2d22b92771935p-4, 0x1.75ecf26aba06dp-3,
        0x1.dd05b486a1932p-5, 0x1.881d75af971d5p-4, 0x1.aee52831aee0cp-5, 0x1.13f57ad1b1befp-53, 0x1.24ff9c8e09330p-3, 0x1.0400000000000p+0, -0x1.5b8b38a68699bp-8, 0x1.2c00000000000p-3,
        0x1.02caa59374e09p+0, 0x1.35ebed44e8beap-4, 0x1.77d1a92e4be8ap-3, 0x1.ec7064a6c34fdp-5, 0x1.8d41e972f6e07p-4, 0x1.beccdf9845f69p-5, 0x1.ba1fa945c4185p-55, 0x1.2d15583c058b0p-3,
        0x1.0400000000000p+0, -0x1.355a6c8b1f774p-8, 0x1.3400000000000p-3, 0x1.02f1f03dc7745p+0, 0x1.3ec0ac1ee9f61p-4, 0x1.79c5e4a82e6d2p-3, 0x1.fc0f719b1ef72p-5, 0x1.9296a2aa943e5p-4,
        0x1.cf141ef8b9de7p-5, -0x1.34081083c8716p-55, 0x1.352c49d6e5610p-3, 0x1.0400000000000p+0, -0x1.0e0fc2388bb30p-8, 0x1.3c00000000000p-3, 0x1.031a563d81251p+0, 0x1.47a15370b721fp-4,
        0x1.7bc9da28731e5p-3, 0x1.05f261e305be9p-4, 0x1.981cc5fa50fbdp-4, 0x1.dfbef42ac4083p-5, 0x1.20acba8e107c7p-53, 0x1.3d447a336f5e0p-3, 0x1.0400000000000p+0, -0x1.cb5384fdb5d5cp-9,
        0x1.4400000000000p-3, 0x1.0343d9159d86fp+0, 0x1.508e423b3747cp-4, 0x1.7dddc0ed597cbp-3, 0x1.0df9279adf104p-4, 0x1.9dd584658d945p-4, 0x1.f0d1914aca06bp-5, -0x1.4e10ddf636efep-53,
        0x1.455df23252c00p-3, 0x1.0400000000000p+0, -0x1.784dd4c4f221ap-9, 0x1.4c00000000000p-3, 0x1.036e7a550d410p+0, 0x1.5987d8d0af1e7p-4, 0x1.8001d22f39726p-3, 0x1.161d0a1116d73p-4,
        0x1.a3c21bbea1528p-4, 0x1.0128274202ff6p-4, 0x1.a0611d10866e2p-53, 0x1.4d78bac086560p-3, 0x1.0400000000000p+0, -0x1.230b55e57df8ap-9, 0x1.5400000000000p-3, 0x1.039a3b96e0f8ap+0,
        0x1.628e78e0c29f6p-4, 0x1.8236492cede01p-3, 0x1.1e5f0fb7b5d84p-4, 0x1.a9e3d71bd08eep-4, 0x1.0a1fd5f7ffab4p-4, -0x1.0f980ef04f6e7p-54, 0x1.5594dcd7a8dc0p-3, 0x1.0400000000000p+0,
        -0x1.9711a47c1d879p-10, 0x1.5c00000000000p-3, 0x1.03c71e8275c12p+0, 0x1.6ba28584c2a23p-4, 0x1.847b6338c3d4bp-3, 0x1.26c0459a55dd8p-4, 0x1.b03c0f5202d6ap-4, 0x1.135229c6466a4p-4,
        0x1.83c9a2a268973p-54, 0x1.5db2617e62a20p-3, 0x1.0400000000000p+0, -0x1.c70bec51f7008p-11, 0x1.6400000000000p-3, 0x1.03f524cba31a9p+0, 0x1.74c4634c49adep-4, 0x1.86d15fc5f33ccp-3,
        0x1.2f41bfa419d7cp-4, 0x1.b6cc2b757e82ap-4, 0x1.1cc18da4d5c39p-4, -0x1.862d42dfb224dp-53, 0x1.65d151c8c8af0p-3, 0x1.0400000000000p+0, -0x1.5b668b9cadbbfp-13, 0x1.6c00000000000p-3,
        0x1.04245032ea88fp+0, 0x1.7df4784a2b473p-4, 0x1.89388076a60f5p-3, 0x1.37e498e8394c1p-4, 0x1.bd95a160f3472p-4, 0x1.2670839844810p-4, 0x1.94228698bc8eap-54, 0x1.6df1b6d8c14a0p-3,
        0x1.0400000000000p+0, 0x1.22819754477acp-11, 0x1.7400000000000p-3, 0x1.0454a285a8cebp+0, 0x1.87332c21b9224p-4, 0x1.8bb1092a93402p-3, 0x1.40a9f3ed3f586p-4, 0x1.c499f643217c8p-4,
        0x1.3061a5d29a16bp-4, -0x1.3b2df3df9f2d7p-53, 0x1.761399de6a160p-3, 0x1.0400000000000p+0, 0x1.528a16a33ab4bp-10, 0x1.7c00000000000p-3, 0x1.04861d9e48d58p+0, 0x1.9080e81461bf6p-4,
        0x1.8e3b400e32ffap-3, 0x1.4992fafb1f2a5p-4, 0x1.cbdabf33705d5p-4, 0x1.3a97a7e23ee89p-4, 0x1.aad12cce44c41p-56, 0x1.7e3704187fae0p-3, 0x1.0400000000000p+0, 0x1.0c3b3c91aaf11p-9,
        0x1.8400000000000p-3, 0x1.04b8c36478509p+0, 0x1.99de170fac1b4p-4, 0x1.90d76daa92166p-3, 0x1.52a0e06c416a6p-4, 0x1.d359a1cdca344p-4, 0x1.451557efd4ca0p-4, 0x1.96ca535a8895dp-60,
        0x1.865bfed4c6ef0p-3, 0x1.0400000000000p+0, 0x1.7186c8f0a11a4p-9, 0x1.8c00000000000p-3, 0x1.04ec95cd5e248p+0, 0x1.a34b25bb94403p-4, 0x1.9385dcf5ca73ap-3, 0x1.5bd4df01afdbep-4,
        0x1.db1854d61a7a9p-4, 0x1.4fdda00bd47cfp-4, -0x1.d1119727e8b64p-54, 0x1.8e82937077e20p-3, 0x1.0400000000000p+0, 0x1.d92b9abc490cbp-9, 0x1.9400000000000p-3, 0x1.052196dbd2a10p+0,
        0x1.acc882894caa1p-4, 0x1.9646db6427516p-3, 0x1.65303a3a864d7p-4, 0x1.e318a0e3cf3d4p-4, 0x1.5af3878cda678p-4, 0x1.3841dda9d51dfp-53, 0x1.96aacb58aa660p-3, 0x1.0400000000000p+0,
        0x1.2196dbd2a1052p-8, 0x1.9c00000000000p-3, 0x1.0557c8a099990p+0, 0x1.b6569dc268965p-4, 0x1.991ab8f9fba21p-3, 0x1.6eb43eaed1e85p-4, 0x1.eb5c6115c4c63p-4, 0x1.665a347f9aefap-4,
        -0x1.1f8fd03ab3673p-53, 0x1.9ed4b00ac4a60p-3, 0x1.0400000000000p+0, 0x1.57c8a0999905bp-8, 0x1.a400000000000p-3, 0x1.058f2d3a9e674p+0, 0x1.bff5e99873832p-4, 0x1.9c01c85e31ce9p-3,
        0x1.7862426e09ff2p-4, 0x1.f3e583cf0885cp-4, 0x1.7214ed2986239p-4, 0x1.7e3e53e594694p-54, 0x1.a7004b14eb5d0p-3, 0x1.0400000000000p+0, 0x1.8f2d3a9e6746bp-8, 0x1.ac00000000000p-3,
        0x1.05c7c6d731ecbp+0, 0x1.c9a6da34fa4b3p-4, 0x1.9efc5eed9c253p-3, 0x1.823ba5614faebp-4, 0x1.fcb60b7ce698fp-4, 0x1.7e27199f3292fp-4, -0x1.842c6068d709cp-54, 0x1.af2da61674110p-3,
        0x1.0400000000000p+0, 0x1.c7c6d731ecae2p-8, 0x1.b400000000000p-3, 0x1.060197b24a973p+0, 0x1.d369e5ca0a798p-4, 0x1.a20ad4cf0db64p-3, 0x1.8c41d1b1a3f31p-4, 0x1.02e807b35e049p-3,
        0x1.8a94456fb8a97p-4, -0x1.cbf9cd337b37cp-53, 0x1.b75ccac059370p-3, 0x1.0800000000000p+0, -0x1.fe684db568d78p-8, 0x1.bc00000000000p-3, 0x1.063ca216c6801p+0, 0x1.dd3f84a32c9fdp-4,
        0x1.a52d850843bc9p-3, 0x1.96763c324648ap-4, 0x1.079ade4407899p-3, 0x1.976021663a5dcp-4, -0x1.3adc3c637289dp-53, 0x1.bf8dc2d5b06a0p-3, 0x1.0800000000000p+0, -0x1.c35de9397fea6p-8,
        0x1.c400000000000p-3, 0x1.0678e85eafb1fp+0, 0x1.e7283136deac6p-4, 0x1.a864cd93a817dp-3, 0x1.a0da64cf7089bp-4, 0x1.0c74ab3abb322p-3, 0x1.a48e8562e6e1ep-4, -0x1.51e3e7eb8fff8p-54,
        0x1.c7c0982c22b80p-3, 0x1.0800000000000p+0, -0x1.8717a1504e0d3p-8, 0x1.cc00000000000p-3, 0x1.06b66cf382a59p+0, 0x1.f1246838936fbp-4, 0x1.abb10f76f5c94p-3, 0x1.ab6fd701a77aep-4,
        0x1.11769c26702c6p-3, 0x1.b223724cdf38ep-4, -0x1.db69ae28307a9p-55, 0x1.cff554ac67190p-3, 0x1.0800000000000p+0, -0x1.49930c7d5a6b9p-8, 0x1.d400000000000p-3, 0x1.06f5324e7707fp+0,
        0x1.fb34a8ab3cbb2p-4, 0x1.af12aedac8d74p-3, 0x1.b6382a45da614p-4, 0x1.16a1ead8e9f44p-3, 0x1.c023141e7749dp-4, 0x1.6ca2722dc16a2p-56, 0x1.d82c0252bf240p-3, 0x1.0800000000000p+0,
        -0x1.0acdb188f814bp-8, 0x1.dc00000000000p-3, 0x1.07353af8cada0p+0, 0x1.02acb9fa32dc9p-3, 0x1.b28a132323718p-3, 0x1.c135029a8f15ep-4, 0x1.1bf7ddeb270e1p-3, 0x1.ce91c40d67463p-4,
        0x1.6e976104baa08p-53, 0x1.e064ab2f76140p-3, 0x1.0800000000000p+0, -0x1.958a0e6a4bfc9p-9, 0x1.e400000000000p-3, 0x1.0776898c0ffd9p+0, 0x1.07c9a6f7f1af0p-3, 0x1.b617a708f2afbp-3,
        0x1.cc6811025b50cp-4, 0x1.2179c9487453ap-3, 0x1.dd740ad09b3abp-4, -0x1.d32db189038c0p-55, 0x1.e89f596762300p-3, 0x1.0800000000000p+0, -0x1.12ece7e004d50p-9, 0x1.ec00000000000p-3,
        0x1.07b920b27c417p+0, 0x1.0cf15e821087ap-3, 0x1.b9bbd8b49dc8cp-3, 0x1.d7d3140bef5c2p-4, 0x1.27290ec080575p-3, 0x1.eccea3056a6a9p-4, 0x1.de5060c9b27a2p-54, 0x1.f0dc173468a50p-3,
        0x1.0800000000000p+0, -0x1.1b7d360efa34dp-10, 0x1.f400000000000p-3, 0x1.07fd03273c018p+0, 0x1.1224251b87f08p-3, 0x1.bd7719d9ab2bcp-3, 0x1.e377d85ffa125p-4, 0x1.2d071ea0cfe55p-3,
        0x1.fca67bb61ddd3p-4, -0x1.2538388a645e7p-53, 0x1.f91aeee603f40p-3, 0x1.0800000000000p+0, -0x1.7e6c61ff422b6p-15, 0x1.fc00000000000p-3, 0x1.084233b6c76f2p+0, 0x1.176240a1df897p-3,
        0x1.c149dfd38779dp-3, 0x1.ef58395531ecdp-4, 0x1.33157855fa966p-3, 0x1.06805d81e6baap-3, 0x1.6827e1b47faecp-55, 0x1.00adf570e6798p-2, 0x1.0800000000000p+0, 0x1.08cedb1dbc656p-10,
        0x1.0200000000000p-2, 0x1.0888b53f3a97bp+0, 0x1.1cabf858525d6p-3, 0x1.c534a3c37af90p-3, 0x1.fb76218ad312ap-4, 0x1.3955ab151caadp-3, 0x1.0ef1607ade82dp-3, 0x1.eef44fcde8746p-53,
        0x1.04cf8ad203480p-2, 0x1.0800000000000p+0, 0x1.116a7e752f5a1p-9, 0x1.0600000000000p-2, 0x1.08d08ab0b03f8p+0, 0x1.220194f34eee8p-3, 0x1.c937e2afdabdep-3, 0x1.03e9c5c4f35bap-3,
        0x1.3fc9568df21a6p-3, 0x1.17a9153843c52p-3, -0x1.d6f54c2bb835ap-54, 0x1.08f23ce0162b8p-2, 0x1.0800000000000p+0, 0x1.a11561607ef23p-9, 0x1.0a00000000000p-2, 0x1.0919b70d9fa87p+0,
        0x1.276360a456c09p-3, 0x1.cd541da483778p-3, 0x1.0a394136d6630p-3, 0x1.46722ba615e9cp-3, 0x1.20aa6a2bc6f73p-3, 0x1.9d0067f1d9d86p-53, 0x1.0d1610f0c1ec8p-2, 0x1.0800000000000p+0,
        0x1.19b70d9fa8688p-8, 0x1.0e00000000000p-2, 0x1.09643d6b3d5d1p+0, 0x1.2cd1a72641546p-3, 0x1.d189d9d4ac7ecp-3, 0x1.10aa9149c2e66p-3, 0x1.4d51ed3de8741p-3, 0x1.29f86f6da4768p-3,
        0x1.ea900828c2a81p-53, 0x1.113b0c65d88c8p-2, 0x1.0800000000000p+0, 0x1.643d6b3d5d119p-8, 0x1.1200000000000p-2, 0x1.09b020f1df195p+0, 0x1.324cb5c9e6b3fp-3, 0x1.d5d9a0be228b7p-3,
        0x1.173ecd29602b0p-3, 0x1.546a70ffa7799p-3, 0x1.3396587ba569fp-3, -0x1.e32589956f2c3p-53, 0x1.156134ada6ff0p-2, 0x1.0800000000000p+0, 0x1.b020f1df1952fp-8, 0x1.1600000000000p-2,
        0x1.09fd64dd62eb0p+0, 0x1.37d4db8335de5p-3, 0x1.da44004dfa3f1p-3, 0x1.1df7155e59412p-3, 0x1.5bbda0394b72ep-3, 0x1.3d877e1177398p-3, 0x1.8ac883b5720a7p-53, 0x1.19888f43427a8p-2,
        0x1.0800000000000p+0, 0x1.fd64dd62eaf85p-8, 0x1.1a00000000000p-2, 0x1.0a4c0c7d99a5fp+0, 0x1.3d6a68f6bb942p-3, 0x1.dec98b06cb8a9p-3, 0x1.24d49432c74b1p-3, 0x1.634d78c1c6ec6p-3,
        0x1.47cf601bf2560p-3, 0x1.3ede7476e25c7p-53, 0x1.1db121aed7720p-2, 0x1.0c00000000000p+0, -0x1.b3f382665a126p-8, 0x1.1e00000000000p-2, 0x1.0a9c1b36b4c8bp+0, 0x1.430db0879e39bp-3,
        0x1.e36ad82887d8bp-3, 0x1.2bd87e1b33c79p-3, 0x1.6b1c0dea4e95ep-3, 0x1.5271a7c90504ap-3, 0x1.afad98a6ebd08p-53, 0x1.21daf185fa360p-2, 0x1.0c00000000000p+0, -0x1.63e4c94b3751cp-8,
        0x1.2200000000000p-2, 0x1.0aed9481b7eedp+0, 0x1.48bf066613bb3p-3, 0x1.e82883d9fdd8fp-3, 0x1.3304122470bf2p-3, 0x1.732b897c5b476p-3, 0x1.5d7229b614f73p-3, 0x1.96b82759745c8p-53,
        0x1.2606046bf95b8p-2, 0x1.0c00000000000p+0, -0x1.126b7e48112dcp-8, 0x1.2600000000000p-2, 0x1.0b407becedf0ep+0, 0x1.4e7ec09e5699ap-3, 0x1.ed032f541ec2dp-3, 0x1.3a589a6688484p-3,
        0x1.7b7e2cc5228bdp-3, 0x1.68d4e83ecad1fp-3, 0x1.985867cb79363p-53, 0x1.2a32601231ec8p-2, 0x1.0c00000000000p+0, -0x1.7f0826241e449p-9, 0x1.2a00000000000p-2, 0x1.0b94d51c61d1cp+0,
        0x1.544d37281f837p-3, 0x1.f1fb810f19f89p-3, 0x1.41d76c7d08a44p-3, 0x1.841651af4e5e6p-3, 0x1.749e15ee5d838p-3, 0x1.a2a36a1f9a890p-55, 0x1.2e600a3865760p-2, 0x1.0c00000000000p+0,
        -0x1.acab8e78b8e2fp-10, 0x1.2e00000000000p-2, 0x1.0beaa3ca5b9c1p+0, 0x1.5a2ac3f6a91d3p-3, 0x1.f71224f1650dbp-3, 0x1.4981ea04f63e7p-3, 0x1.8cf66bebc9b64p-3, 0x1.80d2181598bf7p-3,
        -0x1.841438e0fd320p-54, 0x1.328f08ad12008p-2, 0x1.0c00000000000p+0, -0x1.55c35a463f3fdp-12, 0x1.3200000000000p-2, 0x1.0c41ebc7e151ap+0, 0x1.6017c30943e19p-3, 0x1.fc47cc80c760dp-3,
        0x1.51598120b129dp-3, 0x1.96210a2a855b5p-3, 0x1.8d7589880230dp-3, -0x1.4d129bf178596p-53, 0x1.36bf614dcc050p-2, 0x1.0c00000000000p+0, 0x1.07af1f854661ep-10, 0x1.3600000000000p-2,
        0x1.0c9ab0fd3c135p+0, 0x1.6614927c80482p-3, 0x1.00ce978ac0dddp-2, 0x1.595fad02204b1p-3, 0x1.9f98d7642750dp-3, 0x1.9a8d3d82ac48ap-3, 0x1.77587289b3951p-54, 0x1.3af11a079a6d8p-2,
        0x1.0c00000000000p+0, 0x1.3561fa7826a0dp-9, 0x1.3a00000000000p-2, 0x1.0cf4f76a81a69p+0, 0x1.6c21929bf5acdp-3, 0x1.03898507d5dd4p-2, 0x1.6195f67b79439p-3, 0x1.a9609c35a709fp-3,
        0x1.a81e42bf7455cp-3, 0x1.03304f424551ep-53, 0x1.3f2438d754b40p-2, 0x1.0c00000000000p+0, 0x1.e9eed5034d2a8p-9, 0x1.3e00000000000p-2, 0x1.0d50c3282280dp+0, 0x1.723f25f4acc23p-3,
        0x1.0655108771131p-2, 0x1.69fdf4970163ep-3, 0x1.b37b404ee9a0ap-3, 0x1.b62de6b79bc18p-3, 0x1.ecf2502a2f456p-53, 0x1.4358c3ca032e0p-2, 0x1.0c00000000000p+0, 0x1.50c3282280d28p-8,
        0x1.4200000000000p-2, 0x1.0dae18677c82dp+0, 0x1.786db16834abep-3, 0x1.09319f1631731p-2, 0x1.72994d36297afp-3, 0x1.bdebcbf583888p-3, 0x1.c4c1b918e2ae6p-3, 0x1.92f70f34a155cp-53,
        0x1.478ec0fd419c8p-2, 0x1.0c00000000000p+0, 0x1.ae18677c82d53p-8, 0x1.4600000000000p-2, 0x1.0e0cfb73728f8p+0, 0x1.7ead9c406a36ap-3, 0x1.0c1f991bda616p-2, 0x1.7b69b5b86c42bp-3,
        0x1.c8b5699cd8c9fp-3, 0x1.d3df8f7084936p-3, -0x1.23b7454942387p-54, 0x1.4bc6369fa40e8p-2, 0x1.1000000000000p+0, -0x1.f3048c8d707bbp-8, 0x1.4a00000000000p-2, 0x1.0e6d70b1092b8p+0,
        0x1.84ff5043f9011p-3, 0x1.0f1f6a7afd6ebp-2, 0x1.8470f3aa5d7b9p-3, 0x1.d3db6794e9cfdp-3, 0x1.e38d890fb69fdp-3, 0x1.cfa2dc2327dc5p-54, 0x1.4fff2af11e2c0p-2, 0x1.1000000000000p+0,
        -0x1.928f4ef6d4848p-8, 0x1.4e00000000000p-2, 0x1.0ecf7ca008550p+0, 0x1.8b6339cb9eca7p-3, 0x1.123182b20ac3dp-2, 0x1.8db0dd7d5e860p-3, 0x1.df6139d1315afp-3, 0x1.f3d2132d8bc6fp-3,
        0x1.c6a3692e48eeep-54, 0x1.5439a4436d008p-2, 0x1.1000000000000p+0, -0x1.30835ff7aaf92p-8, 0x1.5200000000000p-2, 0x1.0f3323dba2c62p+0, 0x1.91d9c7d83983cp-3, 0x1.155654fdda02ep-2,
        0x1.972b5b48747c7p-3, 0x1.eb4a7bc9105f9p-3, 0x1.0259f6a535ecfp-2, 0x1.7eb36f6ea55c1p-55, 0x1.5875a8fa83538p-2, 0x1.1000000000000p+0, -0x1.99b848ba73c6ap-9, 0x1.5600000000000p-2,
        0x1.0f986b1b22d42p+0, 0x1.98636c29a92edp-3, 0x1.188e587dbe62dp-2, 0x1.a0e26792c37ebp-3, 0x1.f79af2735e8cdp-3, 0x1.0b1d16eccd4c0p-2, 0x1.502b5beae0510p-54, 0x1.5cb33f8cf8ac0p-2,
        0x1.1000000000000p+0, -0x1.9e539374af74cp-10, 0x1.5a00000000000p-2, 0x1.0fff57329d23ap+0, 0x1.9f009b568f082p-3, 0x1.1bda085939db2p-2, 0x1.aad810283b18ap-3, 0x1.022b472f69148p-2,
        0x1.1436239ad0b79p-2, -0x1.d796880828b86p-53, 0x1.60f26e847b130p-2, 0x1.1000000000000p+0, -0x1.519ac5b8bc081p-17, 0x1.5e00000000000p-2, 0x1.1067ed13a9687p+0, 0x1.a5b1cce4f3f61p-3,
        0x1.1f39e3e764545p-2, 0x1.b50e76f90871bp-3, 0x1.08c0b6f487f97p-2, 0x1.1da9067265c20p-2, 0x1.e5b02995723adp-53, 0x1.65333c7e43aa0p-2, 0x1.1000000000000p+0, 0x1.9fb44ea5a1d64p-10,
        0x1.6200000000000p-2, 0x1.10d231ce216d9p+0, 0x1.ac777b63e0b53p-3, 0x1.22ae6ed81d055p-2, 0x1.bf87d3046c5acp-3, 0x1.0f8fefcb29fe4p-2, 0x1.2779dc99a404ep-2, -0x1.2af1ec3202ae8p-53,
        0x1.6975b02b8e378p-2, 0x1.1000000000000p+0, 0x1.a4639c42db2abp-9, 0x1.6600000000000p-2, 0x1.113e2a90e6a24p+0, 0x1.b3522485f2c6bp-3, 0x1.2638315f1d6ccp-2, 0x1.ca46714f9d555p-3,
        0x1.169b3245e397ep-2, 0x1.31acf99479cf7p-2, -0x1.30d3f8992c228p-56, 0x1.6db9d05213b28p-2, 0x1.1000000000000p+0, 0x1.3e2a90e6a2469p-8, 0x1.6a00000000000p-2, 0x1.11abdcaaae6d5p+0,
        0x1.ba42493cf9b23p-3, 0x1.29d7b86106ad4p-2, 0x1.d54cb5e96870bp-3, 0x1.1de4d9975d46dp-2, 0x1.3c46ea709f8a4p-2, -0x1.cb630457b6f5cp-54, 0x1.71ffa3cc87fc8p-2, 0x1.1000000000000p+0,
        0x1.abdcaaae6d53dp-8, 0x1.6e00000000000p-2, 0x1.121b4d8ad589ep+0, 0x1.c1486dd6a8c89p-3, 0x1.2d8d95a283891p-2, 0x1.e09d1cfb4f5a1p-3, 0x1.256f5cf594bb6p-2, 0x1.474c792614c29p-2,
        -0x1.8fb31533051e9p-55, 0x1.7647318b1ad28p-2, 0x1.1400000000000p+0, -0x1.e4b2752a761d6p-8, 0x1.7200000000000p-2, 0x1.128c82c23ab4ap+0, 0x1.c8651a1a6a356p-3, 0x1.315a5ff99abe3p-2,
        0x1.ec3a3be8ee4c2p-3, 0x1.2d3d511207d5dp-2, 0x1.52c2b02fe6df8p-2, -0x1.3f304fe4d8df3p-53, 0x1.7a908093fc1f0p-2, 0x1.1400000000000p+0, -0x1.737d3dc54b622p-8, 0x1.7600000000000p-2,
        0x1.12ff820420f33p+0, 0x1.cf98d96860d89p-3, 0x1.353eb3814f292p-2, 0x1.f826c27e81bf7p-3, 0x1.355169a827352p-2, 0x1.5eaede614c6dfp-2, 0x1.0aedb36c1700cp-55, 0x1.7edb9803e3c28p-2,
        0x1.1400000000000p+0, -0x1.007dfbdf0ccc6p-8, 0x1.7a00000000000p-2, 0x1.1374512719c3ap+0, 0x1.d6e43ad9a717fp-3, 0x1.393b31cfacd12p-2, 0x1.0232be17b8f05p-2, 0x1.3dae7b23873bcp-2,
        0x1.6b169afb712e5p-2, 0x1.94c0c0bc74599p-54, 0x1.83287f0e9cf80p-2, 0x1.1400000000000p+0, -0x1.175db1cc78ca5p-9, 0x1.7e00000000000p-2, 0x1.13eaf625f7844p+0, 0x1.de47d161d9978p-3,
        0x1.3d50822e63dcap-2, 0x1.087ca8b2ec7ebp-2, 0x1.46577c5f619c8p-2, 0x1.77ffca08a73dep-2, 0x1.1dbde6e7b547fp-53, 0x1.87773cff956f8p-2, 0x1.1400000000000p+0, -0x1.509da087bc752p-12,
        0x1.8200000000000p-2, 0x1.14637720c869dp+0, 0x1.e5c433f1fd940p-3, 0x1.417f51d614654p-2, 0x1.0ef2a472052edp-2, 0x1.4f4f888116da6p-2, 0x1.8570a102117b6p-2, -0x1.b6e89214a7328p-53,
        0x1.8bc7d93a70458p-2, 0x1.1400000000000p+0, 0x1.8ddc8321a7479p-10, 0x1.8600000000000p-2, 0x1.14ddda5dda5c4p+0, 0x1.ed59fd9cd3739p-3, 0x1.45c8542c70412p-2, 0x1.1596449c983a8p-2,
        0x1.5899e0ef7ed0bp-2, 0x1.936fabc543499p-2, 0x1.ff50d7b29f22ep-53, 0x1.901a5b3b9cf50p-2, 0x1.1400000000000p+0, 0x1.bbb4bbb4b87e0p-9, 0x1.8a00000000000p-2, 0x1.155a264ac8172p+0,
        0x1.f509cdbca7047p-3, 0x1.4a2c43055a16fp-2, 0x1.1c692d25160c7p-2, 0x1.6239ef68f9906p-2, 0x1.a203d1dfc2ee2p-2, 0x1.d2019671ef39fp-53, 0x1.946eca98f2718p-2, 0x1.1400000000000p+0,
        0x1.5a264ac8171a9p-8, 0x1.8e00000000000p-2, 0x1.15d8617d8ff02p+0, 0x1.fcd4481aafd5ep-3, 0x1.4eabdee72b776p-2, 0x1.236d1377f943fp-2, 0x1.6c33483a56db7p-2, 0x1.b1345c36d6c50p-2,
        -0x1.841e5761537bbp-56, 0x1.98c52f024e808p-2, 0x1.1400000000000p+0, 0x1.d8617d8ff01dep-8, 0x1.9200000000000p-2, 0x1.165892b5b4a9ap+0, 0x1.025d0a8c0a8c6p-2, 0x1.5347ef524e4b6p-2,
        0x1.2aa3bf565edbdp-2, 0x1.7689ac98d2842p-2, 0x1.c108fb128b4ddp-2, -0x1.a5eeb4452a669p-55, 0x1.9d1d904239878p-2, 0x1.1800000000000p+0, -0x1.a76d4a4b56661p-8, 0x1.9600000000000p-2,
        0x1.16dac0dd68bc8p+0, 0x1.065df0ec54c3ap-2, 0x1.5801430c58a12p-2, 0x1.320f0bbcbccefp-2, 0x1.81410d218f380p-2, 0x1.d189cc9371d29p-2, 0x1.8c3c11d6e6ec7p-58, 0x1.a177f63e8ef18p-2,
        0x1.1800000000000p+0, -0x1.253f229743866p-8, 0x1.9a00000000000p-2, 0x1.175ef30ac48a8p+0, 0x1.0a6d3037ba7c0p-2, 0x1.5cd8b06edcd18p-2, 0x1.39b0e7d679188p-2, 0x1.8c5d8c8128143p-2,
        0x1.e2bf639b3613ap-2, -0x1.74080c70c9c76p-55, 0x1.a5d468f92a560p-2, 0x1.1800000000000p+0, -0x1.4219ea76eb06ep-9, 0x1.9e00000000000p-2, 0x1.17e5308107eefp+0, 0x1.0e8b240691386p-2,
        0x1.61cf15ba2319ap-2, 0x1.418b57ff30656p-2, 0x1.97e3824624146p-2, 0x1.f4b2cf30d6589p-2, -0x1.d4ad974dd0c9bp-55, 0x1.aa32f090998f8p-2, 0x1.1800000000000p+0, -0x1.acf7ef81116bcp-12,
        0x1.a200000000000p-2, 0x1.186d80b1e7a9dp+0, 0x1.12b82a98356f0p-2, 0x1.66e5596c051d8p-2, 0x1.49a076d28a49dp-2, 0x1.a3d77de14d616p-2, 0x1.03b6d13502f53p-1, 0x1.517004ad59707p-53,
        0x1.ae939540d3f08p-2, 0x1.1800000000000p+0, 0x1.b602c79ea752fp-10, 0x1.a600000000000p-2, 0x1.18f7eb3ee7285p+0, 0x1.16f4a4ec4af40p-2, 0x1.6c1c6a9b275fdp-2, 0x1.51f2764b886b9p-2,
        0x1.b03e49d72a144p-2, 0x1.0d7cfe7207dd5p-1, -0x1.ace1e8e77d1b2p-53, 0x1.b2f65f63f6c78p-2, 0x1.1800000000000p+0, 0x1.efd67dce509f5p-9, 0x1.aa00000000000p-2, 0x1.198477fabf325p+0,
        0x1.1b40f6dd15edbp-2, 0x1.71754156d090dp-2, 0x1.5a83a0f44ee42p-2, 0x1.bd1cef26149ccp-2, 0x1.17b149ebb7d53p-1, 0x1.18867054c177ap-53, 0x1.b75b5773075f8p-2, 0x1.1800000000000p+0,
        0x1.8477fabf3257bp-8, 0x1.ae00000000000p-2, 0x1.1a132eead20e6p+0, 0x1.1f9d873afa8f4p-2, 0x1.76f0df0ba2b44p-2, 0x1.63565b2776412p-2, 0x1.ca78b8e4b8181p-2, 0x1.22595de92725ap-1,
        -0x1.bda45225ee470p-53, 0x1.bbc28606babe0p-2, 0x1.1c00000000000p+0, -0x1.ecd1152df1a7ep-8, 0x1.b200000000000p-2, 0x1.1aa41848adb16p+0, 0x1.240abfe932abbp-2, 0x1.7c904eed7e85dp-2,
        0x1.6c6d24640b1b3p-2, 0x1.d857381d01020p-2, 0x1.2d7b39938b939p-1, 0x1.12ecb36d76e02p-53, 0x1.c02bf3d843430p-2, 0x1.1c00000000000p+0, -0x1.5be7b7524ea70p-8, 0x1.b600000000000p-2,
        0x1.1b373c839c9acp+0, 0x1.28890dfbc912dp-2, 0x1.8254a666de3cap-2, 0x1.75ca98b57457cp-2, 0x1.e6be47e7e55fep-2, 0x1.391d368ec3777p-1, -0x1.f7efe4d8a80a5p-54, 0x1.c497a9c2247a0p-2,
        0x1.1c00000000000p+0, -0x1.9186f8c6ca8a7p-9, 0x1.ba00000000000p-2, 0x1.1bcca44246029p+0, 0x1.2d18e1d6eb966p-2, 0x1.883f058df9e20p-2, 0x1.7f7172308ff84p-2, 0x1.f5b411cec1692p-2,
        0x1.45460efae7f7ep-1, -0x1.ca88ac247c281p-53, 0x1.c905b0c10d428p-2, 0x1.1c00000000000p+0, -0x1.9addedcfeb6f6p-11, 0x1.be00000000000p-2, 0x1.1c6458645e0a6p+0, 0x1.31baaf4fa598cp-2,
        0x1.8e5097a00cdbdp-2, 0x1.89648a876efa4p-2, 0x1.029f893bb3ba0p-1, 0x1.51fce3e769492p-1, -0x1.3bd0adac78ba6p-57, 0x1.cd7611f4b8a08p-2, 0x1.1c00000000000p+0, 0x1.91619178298dbp-10,
        0x1.c200000000000p-2, 0x1.1cfe620466a93p+0, 0x1.366eedce16113p-2, 0x1.948a93831a262p-2, 0x1.93a6dcb5336b7p-2, 0x1.0ab30f50362a5p-1, 0x1.5f494440f45e4p-1, -0x1.1b23f79a811b8p-53,
        0x1.d1e8d6a0d56c8p-2, 0x1.1c00000000000p+0, 0x1.fcc408cd52690p-9, 0x1.c600000000000p-2, 0x1.1d9aca798215ap+0, 0x1.3b36187135170p-2, 0x1.9aee3c4e92f90p-2, 0x1.9e3b86c3b0a06p-2,
        0x1.13183439d6983p-1, 0x1.6d333444347eep-1, 0x1.e6687141d7adep-54, 0x1.d65e082df5278p-2, 0x1.1c00000000000p+0, 0x1.9aca798215a4dp-8, 0x1.ca00000000000p-2, 0x1.1e399b59577b1p+0,
        0x1.4010ae343e389p-2, 0x1.a17ce1db4a57bp-2, 0x1.a925cbac8ca27p-2, 0x1.1bd2c29ac5009p-1, 0x1.7bc335806abbep-1, 0x1.9743ad953cbeap-55, 0x1.dad5b02a82420p-2, 0x1.2000000000000p+0,
        -0x1.c664a6a884eafp-8, 0x1.ce00000000000p-2, 0x1.1edade7a0ad1ep+0, 0x1.44ff3215d62d8p-2, 0x1.a837e15b2742ep-2, 0x1.b4691557c3a62p-2, 0x1.24e6b9abecca0p-1, 0x1.8b024f75d3619p-1,
        -0x1.0a42b953c1f21p-57, 0x1.df4fd84bbe168p-2, 0x1.2000000000000p+0, -0x1.252185f52e269p-8, 0x1.d200000000000p-2, 0x1.1f7e9df448be1p+0, 0x1.4a022b4103d45p-2, 0x1.af20a5f90f152p-2,
        0x1.c008f6b992e26p-2, 0x1.2e58507c18f30p-1, 0x1.9afa18dce89c2p-1, -0x1.b90a5e5b4e0ddp-55, 0x1.e3cc8a6ec6ef0p-2, 0x1.2000000000000p+0, -0x1.02c4176e83deep-9, 0x1.d600000000000p-2,
        0x1.2024e42567651p+0, 0x1.4f1a253815e48p-2, 0x1.b638a98189f26p-2, 0x1.cc092e11f7bb9p-2, 0x1.382bf968e1c3cp-1, 0x1.abb4c1a4c4551p-1, -0x1.c384dc65ee1e9p-53, 0x1.e84bd099a6620p-2,
        0x1.2000000000000p+0, 0x1.27212b3b2877ep-11, 0x1.da00000000000p-2, 0x1.20cdbbb19d366p+0, 0x1.5447b00190520p-2, 0x1.bd817514ac3d7p-2, 0x1.d86da7501b24ep-2, 0x1.426665d5dcc91p-1,
        0x1.bd3d1db834bbap-1, -0x1.6289264307fe4p-53, 0x1.eccdb4fc685e0p-2, 0x1.2000000000000p+0, 0x1.9b77633a6cd00p-9, 0x1.de00000000000p-2, 0x1.21792f864eb38p+0, 0x1.598b60573e0cap-2,
        0x1.c4fca1e1d9c05p-2, 0x1.e53a7e9c2fb44p-2, 0x1.4d0c8a26e99afp-1, 0x1.cf9eb09a8a359p-1, -0x1.df861d9afa9e0p-53, 0x1.f15241f23b3f8p-2, 0x1.2000000000000p+0, 0x1.792f864eb3836p-8,
        0x1.e200000000000p-2, 0x1.22274adc744f8p+0, 0x1.5ee5cfd785957p-2, 0x1.ccabd9ee01b3ap-2, 0x1.f274030a7b7b5p-2, 0x1.5823a202e0d0dp-1, 0x1.e2e5b9eebe829p-1, -0x1.3bb42e2ea9787p-54,
        0x1.f5d98202994b8p-2, 0x1.2400000000000p+0, -0x1.d8b5238bb0864p-8, 0x1.e600000000000p-2, 0x1.22d8193b1990ap+0, 0x1.64579d3920d0fp-2, 0x1.d490d8e4fe1fep-2, 0x1.000f5cbd3ed59p-1,
        0x1.63b134e45f774p-1, 0x1.f71f42fd578cep-1, 0x1.8ad1cc0e1ac47p-53, 0x1.fa637fe27bf60p-2, 0x1.2400000000000p+0, -0x1.27e6c4e66f5a1p-8, 0x1.ea00000000000p-2, 0x1.238ba679f6ae1p+0,
        0x1.69e16c815a8f5p-2, 0x1.dcad6cf6cd4c9p-2, 0x1.071fafd2ade38p-1, 0x1.6fbb1afee9630p-1, 0x1.062c96a7acb82p+0, 0x1.e358035d3555bp-56, 0x1.fef0467599588p-2, 0x1.2400000000000p+0,
        -0x1.d166182547b6fp-10, 0x1.ee00000000000p-2, 0x1.2441fec425f4bp+0, 0x1.6f83e73cf67b4p-2, 0x1.e50377c1691bap-2, 0x1.0e6d77af8190ep-1, 0x1.7c47827f29078p-1, 0x1.1151213b5ffdcp+0,
        0x1.cc7a15feba301p-57, 0x1.01bff067d6224p-1, 0x1.2400000000000p+0, 0x1.07fb1097d2bdcp-10, 0x1.f200000000000p-2, 0x1.24fb2e9af6533p+0, 0x1.753fbcbbea804p-2, 0x1.ed94ef480e731p-2,
        0x1.15fb5106d90c6p-1, 0x1.895cf52dad430p-1, 0x1.1d05228faae13p+0, -0x1.50976e849f35ap-53, 0x1.04092d1ae3b48p-1, 0x1.2400000000000p+0, 0x1.f65d35eca665dp-9, 0x1.f600000000000p-2,
        0x1.25b742d8dc7fap+0, 0x1.7b15a25013475p-2, 0x1.f663def8d6387p-2, 0x1.1dcbfa2df4bffp-1, 0x1.97025e7c2e4e5p-1, 0x1.295101c2ae4abp+0, 0x1.4c8dcb02a3d13p-53, 0x1.0653df0fd9fd8p-1,
        0x1.2400000000000p+0, 0x1.b742d8dc7fa40p-8, 0x1.fa00000000000p-2, 0x1.267648b4843f2p+0, 0x1.8106538f10257p-2, 0x1.ff7268c1920b1p-2, 0x1.25e255148d4e4p-1, 0x1.a53f12061c3fep-1,
        0x1.363db5b9300e5p+0, -0x1.47774624b8b97p-53, 0x1.08a00c1cae338p-1, 0x1.2800000000000p+0, -0x1.89b74b7bc0e50p-8, 0x1.fe00000000000p-2, 0x1.27384dc4036f2p+0, 0x1.8712929775c8fp-2,
        0x1.0461631a78776p-1, 0x1.2e41695ee0c65p-1, 0x1.b41ad28e05161p-1, 0x1.43d4cff1df849p+0, 0x1.5941cbabba919p-53, 0x1.0aedba3221c1cp-1, 0x1.2800000000000p+0, -0x1.8f6477f921c27p-9,
        0x1.0100000000000p-1, 0x1.27fd600030888p+0, 0x1.8d3b28598a1b5p-2, 0x1.092ba4e0bc755p-1, 0x1.36ec66a428eecp-1, 0x1.c39c044f514c9p-1, 0x1.5220518c4ef3ap+0, 0x1.0c1d1a852f235p+1,
        0x1.78082d00f64b8p-53, 0x1.0d3cef5c846f8p-1, 0x1.2800000000000p+0, -0x1.4fffe7bbc39dfp-15, 0x1.0300000000000p-1, 0x1.28c58dc81e6d7p+0, 0x1.9380e4e3bf356p-2, 0x1.0e192ffc646a7p-1,
        0x1.3fe6a6d34756dp-1, 0x1.d3cef139abc91p-1, 0x1.612b8f80111c0p+0, 0x1.1a33c3467c688p+1, -0x1.a995434f59445p-55, 0x1.0f8db1c47d550p-1, 0x1.2800000000000p+0, 0x1.8b1b903cdae3fp-9,
        0x1.0500000000000p-1, 0x1.2990e5e4bf713p+0, 0x1.99e49fb326e9ep-2, 0x1.132b48779391ap-1, 0x1.4933b0c2fe325p-1, 0x1.e4bb1aeaae1d0p-1, 0x1.710201f4377bdp+0, 0x1.292711c886605p+1,
        -0x1.33ab17130ce99p-53, 0x1.11e007afdaf10p-1, 0x1.2800000000000p+0, 0x1.90e5e4bf712c7p-8, 0x1.0700000000000p-1, 0x1.2a5f778cb1a3bp+0, 0x1.a06738081c5d1p-2, 0x1.186340d5e6499p-1,
        0x1.52d73aedd6be6p-1, 0x1.f66a51cf1aaa0p-1, 0x1.81b024834e5a9p+0, 0x1.39066fce48906p+1, -0x1.34e4f6bfb4c85p-53, 0x1.1433f7826aad4p-1, 0x1.2c00000000000p+0, -0x1.a088734e5c574p-8,
        0x1.0900000000000p-1, 0x1.2b315268368dbp+0, 0x1.a709953f655b7p-2, 0x1.1dc27ad9032ecp-1, 0x1.5cd52e5f88e23p-1, 0x1.047380a68bdfcp+0, 0x1.934352f057820p+0, 0x1.49e27dae8a2fcp+1,
        0x1.6832cfaa44565p-55, 0x1.168987bed8260p-1, 0x1.2c00000000000p+0, -0x1.9d5b2f92e4a41p-9, 0x1.0b00000000000p-1, 0x1.2c06869558a9ep+0, 0x1.adcca73011b64p-2, 0x1.234a68511146ap-1,
        0x1.6731a9d6cbf3cp-1, 0x1.0e1e1d575f00ap+0, 0x1.a5c9dadea17e7p+0, 0x1.5bcd2d9123e7cp+1, -0x1.23e4fcc2ae1e4p-53, 0x1.18e0bf07948f0p-1, 0x1.2c00000000000p+0, 0x1.a1a5562a7614ap-14,
        0x1.0d00000000000p-1, 0x1.2cdf24ac410c6p+0, 0x1.b4b1668e63d97p-2, 0x1.28fc8bfa256b2p-1, 0x1.71f1051fdf05ap-1, 0x1.183ae0753c882p+0, 0x1.b9530f1921090p+0, 0x1.6ed9e14f942bcp+1,
        0x1.2787989c77fa3p-53, 0x1.1b39a41fc691cp-1, 0x1.2c00000000000p+0, 0x1.be49588218cd6p-9, 0x1.0f00000000000p-1, 0x1.2dbb3dc3bfd25p+0, 0x1.bbb8d55413207p-2, 0x1.2eda7a6792bf1p-1,
        0x1.7d17d4ac4e230p-1, 0x1.22d00aae6cb05p+0, 0x1.cdef5c9028e71p+0, 0x1.831d8b40c626cp+1, 0x1.53fef9873f484p-54, 0x1.1d943dec430c0p-1, 0x1.2c00000000000p+0, 0x1.bb3dc3bfd24a1p-8,
        0x1.1100000000000p-1, 0x1.2e9ae3760a19bp+0, 0x1.c2e3ff2e3e2ebp-2, 0x1.34e5dafe1cd38p-1, 0x1.88aaed6ce0b26p-1, 0x1.2de442c4b06c6p+0, 0x1.e3b06138813d2p+0, 0x1.98aed23fd5612p+1,
        -0x1.1ec19b7af0e54p-54, 0x1.1ff093748f114p-1, 0x1.3000000000000p+0, -0x1.651c89f5e657ep-8, 0x1.1300000000000p-1, 0x1.2f7e27e5b072bp+0, 0x1.ca33f9f169c4dp-2, 0x1.3b2068fe1eb56p-1,
        0x1.94af68f30e1b7p-1, 0x1.397e9cfcf9887p+0, 0x1.faa904fb7f25fp+0, 0x1.afa6394745d90p+1, 0x1.6955c2a139390p-54, 0x1.224eabe3eba20p-1, 0x1.3000000000000p+0, -0x1.03b0349f1aa85p-9,
        0x1.1500000000000p-1, 0x1.30651dc2d0e76p+0, 0x1.d1a9e613ef408p-2, 0x1.418bf49ed083dp-1, 0x1.a12aa9dfd1e23p-1, 0x1.45a6a32b75f76p+0, 0x1.0976ca7673f47p+1, 0x1.c81e4b046ac6ap+1,
        0x1.79ff77d1beb80p-55, 0x1.24ae8e8a6b8b0p-1, 0x1.3000000000000p+0, 0x1.94770b439d90ep-10, 0x1.1700000000000p-1, 0x1.314fd85087ecdp+0, 0x1.d946ef2f45390p-2, 0x1.482a643beda05p-1,
        0x1.ae2260a640dd7p-1, 0x1.52645d6a3d695p+0, 0x1.1649f08098fe0p+1, 0x1.e233c9d2bade7p+1, 0x1.e948c4e5f8348p-54, 0x1.271042de13e58p-1, 0x1.3000000000000p+0, 0x1.4fd85087ecd1ap-8,
        0x1.1900000000000p-1, 0x1.323e6b6aa3c67p+0, 0x1.e10c4c8894828p-2, 0x1.4efdb59718389p-1, 0x1.bb9c90a8d7622p-1, 0x1.5fc05b8a62b12p+0, 0x1.23d9ae4296831p+1, 0x1.fe05e49c0b830p+1,
        0x1.19107c1189de8p-53, 0x1.2973d07c07bccp-1, 0x1.3400000000000p+0, -0x1.c194955c3993dp-8, 0x1.1b00000000000p-1, 0x1.3330eb8b9e20bp+0, 0x1.e8fb41a11468bp-2, 0x1.5607ff2e740e1p-1,
        0x1.c99f95b91db14p-1, 0x1.6dc3bf50c5faap+0, 0x1.3232a0cfac1c7p+1, 0x1.0ddb3894efd30p+2, -0x1.7760f3783d916p-53, 0x1.2bd93f29bf5f0p-1, 0x1.3400000000000p+0, -0x1.9e28e8c3bea7fp-9,
        0x1.1d00000000000p-1, 0x1.34276dd2dfe6dp+0, 0x1.f1151eceb226bp-2, 0x1.5d4b71aa123cep-1, 0x1.d8322a01f65f8p-1, 0x1.7c784791ce583p+0, 0x1.41625c15a6b9cp+1, 0x1.1db5164280febp+2,
        0x1.1046328ca6dbbp-53, 0x1.2e4096d64beacp-1, 0x1.3400000000000p+0, 0x1.3b6e96ff364e1p-11, 0x1.1f00000000000p-1, 0x1.3522080b539c7p+0, 0x1.f95b41dd91a82p-2, 0x1.64ca5961ea9cap-1,
        0x1.e75b6c65b3b2fp-1, 0x1.8be85c412e59fp+0, 0x1.5177802462a51p+1, 0x1.2ea48109fc81bp+2, 0x1.59e4c9f70ca98p-54, 0x1.30a9df9ba7b3cp-1, 0x1.3400000000000p+0, 0x1.22080b539c6a2p-8,
        0x1.2100000000000p-1, 0x1.3620d0b24acdap+0, 0x1.00e78b5d803b6p-1, 0x1.6c871ffe457fbp-1, 0x1.f722e759ef386p-1, 0x1.9c1f1b8d0e874p+0, 0x1.6281d080fb06ep+1, 0x1.40bf2d1f69df7p+2,
        -0x1.489eacbfaf37fp-55, 0x1.331521c0141bcp-1, 0x1.3800000000000p+0, -0x1.df2f4db532667p-8, 0x1.2300000000000p-1, 0x1.3723defebb76dp+0, 0x1.05390c153fc4cp-1, 0x1.74844e34a2666p-1,
        0x1.03c84c260a400p+0, 0x1.ad28681e70f01p+0, 0x1.74924dbc4a78ep+1, 0x1.541cb8bbca2e0p+2, 0x1.c75287bea8472p-54, 0x1.358265b7858f8p-1, 0x1.3800000000000p+0, -0x1.b842028912510p-9,
        0x1.2500000000000p-1, 0x1.382b4ae8da9c7p+0, 0x1.09a2e842cabb9p-1, 0x1.7cc48da356141p-1, 0x1.0c567bcd4fdb8p+0, 0x1.bf10f89b62c32p+0, 0x1.87bb518adc4b9p+1, 0x1.68d6dc516f6f1p+2,
        -0x1.33a6b2e37d6a3p-54, 0x1.37f1b4251e5acp-1, 0x1.3800000000000p+0, 0x1.5a5746d4e3a7ap-11, 0x1.2700000000000p-1, 0x1.39372d3219a4cp+0, 0x1.0e25ed4394437p-1, 0x1.854aaace4cc74p-1,
        0x1.154080981ee13p+0, 0x1.d1e6688aa5332p+0, 0x1.9c10ada3bd18fp+1, 0x1.7f099ffe4ae21p+2, -0x1.ed56b7b588abep-53, 0x1.3a6315dcb911cp-1, 0x1.3800000000000p+0, 0x1.372d3219a4ba9p-8,
        0x1.2900000000000p-1, 0x1.3a479f6d8c6bcp+0, 0x1.12c2f11197a32p-1, 0x1.8e19973f949d5p-1, 0x1.1e8b1ee7a481dp+0, 0x1.e5b74abbe8828p+0, 0x1.b1a7cdb3d83bcp+1, 0x1.96d396dc46100p+2,
        0x1.7798cacd8f69cp-53, 0x1.3cd693e4835e8p-1, 0x1.3c00000000000p+0, -0x1.b860927394391p-8, 0x1.2b00000000000p-1, 0x1.3b5cbc08be738p+0, 0x1.177ad2b5cb6b7p-1, 0x1.97346bce90eb1p-1,
        0x1.283b668ec7f04p+0, 0x1.fa933d5b8ed04p+0, 0x1.c897dcbcdff9ap+1, 0x1.b05620e4abf55p+2, 0x1.81ff61ee42043p-55, 0x1.3f4c3776aa08cp-1, 0x1.3c00000000000p+0, -0x1.4687ee8319086p-9,
        0x1.2d00000000000p-1, 0x1.3c769e54fe05ep+0, 0x1.1c4e7ac1a81a0p-1, 0x1.a09e6b10fa326p-1, 0x1.3256b840f679bp+0, 0x1.08457fee9ef1ap+1, 0x1.e0f9ee4146343p+1, 0x1.cbb5b433496a9p+2,
        -0x1.57a4c59f087c0p-53, 0x1.41c40a03171a8p-1, 0x1.3c00000000000p+0, 0x1.da7953f81773dp-10, 0x1.2f00000000000p-1, 0x1.3d956291249dcp+0, 0x1.213edbd044ac9p-1, 0x1.aa5b03f917fa8p-1,
        0x1.3ce2cb7380a79p+0, 0x1.13d84576afae8p+1, 0x1.fae92baab74f3p+1, 0x1.e91a2e9129e4ap+2, 0x1.056710cec83f7p-54, 0x1.443e153143194p-1, 0x1.3c00000000000p+0, 0x1.956291249dbc4p-8,
        0x1.3100000000000p-1, 0x1.3eb925f3e4715p+0, 0x1.264cf30f965d1p-1, 0x1.b46dd4a4f2fb2p-1, 0x1.47e5b4bc2e94fp+0, 0x1.200b954f8f9ebp+1, 0x1.0b41833305d9fp+2, 0x1.04579826ef167p+3,
        -0x1.737a0e06ebcaep-54, 0x1.46ba62e21a53cp-1, 0x1.4000000000000p+0, -0x1.46da0c1b8eb04p-8, 0x1.3300000000000p-1, 0x1.3fe206b6a38d5p+0, 0x1.2b79c8d26c7a0p-1, 0x1.bedaad62978f8p-1,
        0x1.5365ecb8cc6d1p+0, 0x1.2ce9cd894af54p+1, 0x1.19f3b79f7c63ep+2, 0x1.155240c8e7b9ep+3, 0x1.485d013dc9a80p-53, 0x1.4938fd31f754cp-1, 0x1.4000000000000p+0, -0x1.df9495c72b1e7p-12,
        0x1.3500000000000p-1, 0x1.41102420ed8e7p+0, 0x1.30c6712bce2b2p-1, 0x1.c9a593ede345ep-1, 0x1.5f6a578cab466p+0, 0x1.3a7e13ead62eep+1, 0x1.299c89cb9a228p+2, 0x1.279741be749b0p+3,
        0x1.fe28fc6a9831fp-53, 0x1.4bb9ee7ab3a40p-1, 0x1.4000000000000p+0, 0x1.102420ed8e776p-8, 0x1.3700000000000p-1, 0x1.42439e9485b43p+0, 0x1.36340c946e033p-1, 0x1.d4d2c6ecc5a7ep-1,
        0x1.6bfa4d027255ap+0, 0x1.48d4672504be1p+1, 0x1.3a4ed09445bd5p+2, 0x1.3b435749e19f9p+3, -0x1.0e7e5eaaaf53ep-59, 0x1.4e3d4155d0070p-1, 0x1.4400000000000p+0, -0x1.bc616b7a4bd36p-8,
        0x1.3900000000000p-1, 0x1.437c979a23c23p+0, 0x1.3bc3c89af6a9dp-1, 0x1.e066c1af553bap-1, 0x1.791da1622569ap+0, 0x1.57f9b1b18ae2bp+1, 0x1.4c1f088bff240p+2, 0x1.50761019a4522p+3,
        0x1.4c238fdfccb13p-53, 0x1.50c3009eb58f8p-1, 0x1.4400000000000p+0, -0x1.06d0cbb87b9e1p-9, 0x1.3b00000000000p-1, 0x1.44bb31eee6f35p+0, 0x1.4176e0a004d1dp-1, 0x1.ec6640399fa54p-1,
        0x1.86dcaf0cfd106p+0, 0x1.67fbde8c80e97p+1, 0x1.5f237d9cd2d79p+2, 0x1.67521c8076345p+3, 0x1.ec756b089f7afp-53, 0x1.534b377510d94p-1, 0x1.4400000000000p+0, 0x1.7663ddcde698fp-9,
        0x1.3d00000000000p-1, 0x1.45ff91928b1bep+0, 0x1.474e9e9eb53e9p-1, 0x1.f8d6439db03b6p-1, 0x1.954060f298b87p+0, 0x1.78e9effc72ab6p+1, 0x1.73747941456e7p+2, 0x1.7ffda74a71e71p+3,
        -0x1.eed93fe7483b3p-53, 0x1.55d5f13f48ec0p-1, 0x1.4400000000000p+0, 0x1.ff91928b1bdffp-8, 0x1.3f00000000000p-1, 0x1.4749dbd66d0c4p+0, 0x1.4d4c5c02c2013p-1, 0x1.02de0b56768ccp+0,
        0x1.a4523df53a7bdp+0, 0x1.8ad418a357386p+1, 0x1.892c75e392799p+2, 0x1.9aa2b97746acdp+3, 0x1.24f0ab4a71e44p-54, 0x1.586339ad13548p-1, 0x1.4800000000000p+0, -0x1.6c485325e775ep-9,
        0x1.4100000000000p-1, 0x1.489a376d6c491p+0, 0x1.5371828d40829p-1, 0x1.098ea98450d83p+0, 0x1.b41c755526e3bp+0, 0x1.9dcbd719f540ep+1, 0x1.a0685805d08d1p+2, 0x1.b76faa5142633p+3,
        0x1.ad9a7f1ff56fcp-61, 0x1.5af31cba27244p-1, 0x1.4800000000000p+0, 0x1.346edad892100p-9, 0x1.4300000000000p-1, 0x1.49f0cc7cb94ccp+0, 0x1.59bf8d492aa1ep-1, 0x1.107ff34d2ca82p+0,
        0x1.c4a9ec3df9e51p+0, 0x1.b1e4145f5874ep+1, 0x1.b947adeb92648p+2, 0x1.d6979d903d532p+3, 0x1.4323104c67f5ep-53, 0x1.5d85a6b1109a4p-1, 0x1.4800000000000p+0, 0x1.f0cc7cb94cc1ap-8,
        0x1.4500000000000p-1, 0x1.4b4dc4ada0bf0p+0, 0x1.60380990f861fp-1, 0x1.17b50cbec7542p+0, 0x1.d6064c93cfe8fp+0, 0x1.c731456f36fe3p+1, 0x1.d3ecf696e5374p+2, 0x1.f85311778af1dp+3,
        0x1.a53bf31ebda84p-56, 0x1.601ae42e27660p-1, 0x1.4c00000000000p+0, -0x1.6476a4be81f81p-9, 0x1.4700000000000p-1, 0x1.4cb14b4065600p+0, 0x1.66dc9826ada4fp-1, 0x1.1f314a3298d4dp+0,
        0x1.e83e152191cb4p+0, 0x1.ddc9905ca69afp+1, 0x1.f07df1079c46ap+2, 0x1.0e703f9440eb0p+4, 0x1.495e15817d0ddp-53, 0x1.62b2e222a98a0p-1, 0x1.4c00000000000p+0, 0x1.629680cac00d4p-9,
        0x1.4900000000000p-1, 0x1.4e1b8d203bdc9p+0, 0x1.6daeee5fe0976p-1, 0x1.26f833c44f71ep+0, 0x1.fb5eab4d92f91p+0, 0x1.f5c4f55a779c8p+1, 0x1.0791fa66a7536p+3, 0x1.224287dce5d75p+4,
        0x1.de7e8964f770bp-53, 0x1.654dadd7fd12cp-1, 0x1.5000000000000p+0, -0x1.e472dfc42374bp-8, 0x1.4b00000000000p-1, 0x1.4f8cb8f87d541p+0, 0x1.74b0d767620c4p-1, 0x1.2f0d89126f083p+0,
        0x1.07bb373f08794p+1, 0x1.079ebe1419117p+2, 0x1.18062a917f81ep+3, 0x1.37c6b48444deep+4, 0x1.07a41f4061e08p-54, 0x1.67eb54f31af70p-1, 0x1.5000000000000p+0, -0x1.cd1c1e0aafa85p-10,
        0x1.4d00000000000p-1, 0x1.5104ff4b2718cp+0, 0x1.7be4359659939p-1, 0x1.377545502eae6p+0, 0x1.124a66ae0ac51p+1, 0x1.1527b33524d17p+2, 0x1.29b467fbf7a2dp+3, 0x1.4f274ad716768p+4,
        -0x1.c610f7c204ea8p-54, 0x1.6a8be57825a6cp-1, 0x1.5000000000000p+0, 0x1.04ff4b2718c01p-8, 0x1.4f00000000000p-1, 0x1.52849288c017dp+0, 0x1.834b03e6d3f7fp-1, 0x1.4033a3b0747cbp+0,
        0x1.1d652e946b196p+1, 0x1.238cb3c2f8cb4p+2, 0x1.3cb8053e520c1p+3, 0x1.68938607bc0f6p+4, -0x1.4274ccc053597p-55, 0x1.6d2f6dce2dfb8p-1, 0x1.5400000000000p+0, -0x1.7b6d773fe8364p-8,
        0x1.5100000000000p-1, 0x1.540ba729be713p+0, 0x1.8ae75781f49a2p-1, 0x1.494d2432ac103p+0, 0x1.291479b0015b6p+1, 0x1.32de7156b74e9p+2, 0x1.512f0e8362ec8p+3, 0x1.843fec8d2e0f8p+4,
        -0x1.f55dbbb3acc53p-55, 0x1.6fd5fcc3296f0p-1, 0x1.5400000000000p+0, 0x1.74e537ce2565ep-13, 0x1.5300000000000p-1, 0x1.559a73c98a101p+0, 0x1.92bb616c3163dp-1, 0x1.52c690db2c44dp+0,
        0x1.3561e0f4546b8p+1, 0x1.432f17f099a82p+2, 0x1.673a9831e227ap+3, 0x1.a266fa02bbcd5p+4, 0x1.279a8aea9cb9dp-53, 0x1.727fa1901cb44p-1, 0x1.5400000000000p+0, 0x1.9a73c98a10084p-8,
        0x1.5500000000000p-1, 0x1.573131433b9bdp+0, 0x1.9ac970523e7b2p-1, 0x1.5ca50361f7393p+0, 0x1.4257bb0f40825p+1, 0x1.5492746286025p+2, 0x1.7eff1781495b4p+3, 0x1.c349e0a1139f1p+4,
        -0x1.d2c668b6015dap-58, 0x1.752c6bdd7e0e0p-1, 0x1.5800000000000p+0, -0x1.9d9d7988c865fp-9, 0x1.5700000000000p-1, 0x1.58d01ad039e07p+0, 0x1.a313f279933cdp-1, 0x1.66edeb63d93a6p+0,
        0x1.50012d836441ap+1, 0x1.671e1f23d152cp+2, 0x1.98a4c65d3a1ddp+3, 0x1.e73165ebdbf39p+4, -0x1.e5b6c7aaa4996p-53, 0x1.77dc6bc7d2fa0p-1, 0x1.5800000000000p+0, 0x1.a035a073c0e86p-9,
        0x1.5900000000000p-1, 0x1.5a776e28dadb6p+0, 0x1.ab9d77d7be2b5p-1, 0x1.71a715234c8a9p+0, 0x1.5e6a3f873554cp+1, 0x1.7ae9ac1f33f9bp+2, 0x1.b45814310046ep+3, 0x1.07376f64b03e7p+5,
        -0x1.3f39bb3ab0542p-53, 0x1.7a8fb1e48d158p-1, 0x1.5c00000000000p+0, -0x1.8891d725249cap-8, 0x1.5b00000000000p-1, 0x1.5c276ba730f9bp+0, 0x1.b468b454127c3p-1, 0x1.7cd6b0e816adbp+0,
        0x1.6d9feeddac837p+1, 0x1.900ee0209e3b7p+2, 0x1.d24a257489c7ep+3, 0x1.1caea7f810e14p+5, -0x1.4f20e24f9675bp-55, 0x1.7d464f472a690p-1, 0x1.5c00000000000p+0, 0x1.3b5d3987cd623p-11,
        0x1.5d00000000000p-1, 0x1.5de0566c30cdcp+0, 0x1.bd78823798d1ap-1, 0x1.88835b0e567d8p+0, 0x1.7db046e46660ap+1, 0x1.a6a9eca07caa5p+2, 0x1.f2b1641ecef64p+3, 0x1.34315c36f367bp+5,
        -0x1.08ca1542594a6p-53, 0x1.800055869d9e8p-1, 0x1.5c00000000000p+0, 0x1.e0566c30cdbd9p-8, 0x1.5f00000000000p-1, 0x1.5fa274875fa03p+0, 0x1.c6cfe4cf96d63p-1, 0x1.94b424d7b8313p+0,
        0x1.8eaa7a1b04592p+1, 0x1.bed9b2c5a9d87p+2, 0x1.0ae511bc92f68p+4, 0x1.4df8c685fbd64p+5, -0x1.c07a830fe6378p-53, 0x1.82bdd6c30303cp-1, 0x1.6000000000000p+0, -0x1.762de2817f40cp-10,
        0x1.6100000000000p-1, 0x1.616e0f213fcd6p+0, 0x1.d0720b47784ffp-1, 0x1.a1709e13c6707p+0, 0x1.a09efe70b2e72p+1, 0x1.d8c00e976aad9p+2, 0x1.1debad1ae1ea8p+4, 0x1.6a4536424341fp+5,
        0x1.13e53a65d40b1p-53, 0x1.857ee5aba79e8p-1, 0x1.6000000000000p+0, 0x1.6e0f213fcd614p-8, 0x1.6300000000000p-1, 0x1.634372a8b4ed8p+0, 0x1.da6253bf69915p-1, 0x1.aec0dfb6df86fp+0,
        0x1.b39facaf2d64bp+1, 0x1.f4822b7e2dc06p+2, 0x1.3291bb12537e3p+4, 0x1.895f0af3ef0d1p+5, 0x1.ea58871e7ed76p-53, 0x1.884395856807cp-1, 0x1.6400000000000p+0, -0x1.791aae9624f1cp-9,
        0x1.6500000000000p-1, 0x1.6522ef039f5e3p+0, 0x1.e4a44ea588e54p-1, 0x1.bcad977a3f8a4p+0, 0x1.c7bfe3669f2f2p+1, 0x1.092471aea54a4p+3, 0x1.4900a6b866959p+4, 0x1.ab97d620634cfp+5,
        0x1.48649da91b0fdp-54, 0x1.8b0bfa316d3a0p-1, 0x1.6400000000000p+0, 0x1.22ef039f5e2adp-8, 0x1.6700000000000p-1, 0x1.670cd7c2f4fc3p+0, 0x1.ef3bc2583cf60p-1, 0x1.cb4014a2e1684p+0,
        0x1.dd14adcb9f8fbp+1, 0x1.192094e164373p+3, 0x1.616698fc171bcp+4, 0x1.d14baa46b7be1p+5, -0x1.bac31bbdfe65ap-53, 0x1.8dd828344e08cp-1, 0x1.6800000000000p+0, -0x1.e6507a1607a77p-9,
        0x1.6900000000000p-1, 0x1.6901845aa3c85p+0, 0x1.fa2caf18fbd18p-1, 0x1.da825610c140ep+0, 0x1.f3b4ef08895e1p+1, 0x1.2a4e4272cd203p+3, 0x1.7bf7160c4a0eep+4, 0x1.fae29ec79351dp+5,
        0x1.bf5bb3e22fb0ap-57, 0x1.90a834bd9c858p-1, 0x1.6800000000000p+0, 0x1.01845aa3c8533p-8, 0x1.6b00000000000p-1, 0x1.6b01505d92e4ep+0, 0x1.02bda9abd20d8p+0, 0x1.ea7f19bc5cc13p+0,
        0x1.05dcc94afb2bbp+2, 0x1.3cc8cb382b54ap+3, 0x1.98ebb19c28eaep+4, 0x1.146948f7609b5p+6, 0x1.cd223f66137e5p-54, 0x1.937c35afe73acp-1, 0x1.6c00000000000p+0, -0x1.fd5f44da36334p-9,
        0x1.6d00000000000p-1, 0x1.6d0c9bbe1ef2bp+0, 0x1.0896182fbdd29p+0, 0x1.fb41edcd403ecp+0, 0x1.129ee121d0023p+2, 0x1.50ae5b34159b2p+3, 0x1.b884ddb5ceac4p+4, 0x1.2dd09a0b334b0p+6,
        -0x1.6bf1dd8f14bf9p-54, 0x1.965441a936d24p-1, 0x1.6c00000000000p+0, 0x1.0c9bbe1ef2aeep-8, 0x1.6f00000000000p-1, 0x1.6f23cb13786cfp+0, 0x1.0ea207b7fc134p+0, 0x1.066ba1bd0d518p+1,
        0x1.202f9159ec945p+2, 0x1.6620516ff868ap+3, 0x1.db0ad87398014p+4, 0x1.49f3347d58711p+6, 0x1.d858f54b11a28p-55, 0x1.9930700c1184cp-1, 0x1.7000000000000p+0, -0x1.b869d90f2626ap-9,
        0x1.7100000000000p-1, 0x1.714747e455603p+0, 0x1.14e403a65655fp+0, 0x1.0fa641f4aa7a1p+1, 0x1.2e9f0b946c70ap+2, 0x1.7d43a3cc53936p+3, 0x1.00675ee087279p+5, 0x1.6927877313cefp+6,
        -0x1.b1ba1772d6e62p-53, 0x1.9c10d9090e874p-1, 0x1.7000000000000p+0, 0x1.4747e455602d3p-8, 0x1.7300000000000p-1, 0x1.737780f773decp+0, 0x1.1b5ec1288b243p+0, 0x1.195813a853fa5p+1,
        0x1.3dff06d2743e5p+2, 0x1.9641509b4b924p+3, 0x1.1515e19a59d1fp+5, 0x1.8bd01f3e53877p+6, 0x1.62269fc348baep-54, 0x1.9ef595a90493cp-1, 0x1.7400000000000p+0, -0x1.10fe111842743p-9,
        0x1.7500000000000p-1, 0x1.75b4eaaa78140p+0, 0x1.2215228b49576p+0, 0x1.2388e74d66746p+1, 0x1.4e62ea43083a8p+2, 0x1.b146e02885ed7p+3, 0x1.2bc4529a3bc2cp+5, 0x1.b25d8cdafe7e5p+6,
        0x1.8862df03f8a74p-53, 0x1.a1debfd7dfbd8p-1, 0x1.7400000000000p+0, 0x1.b4eaaa7813fbap-8, 0x1.7700000000000p-1, 0x1.77ffff4fc0008p+0, 0x1.290a3ade499e4p+0, 0x1.2e412ff22fe11p+1,
        0x1.5fdffd7a17943p+2, 0x1.ce86f8af79aefp+3, 0x1.44aca6f8edf86p+5, 0x1.dd50a29cf9f92p+6, 0x1.49db0c5865233p-53, 0x1.a4cc72702bd90p-1, 0x1.7800000000000p+0, -0x1.607fff08268e1p-25,
        0x1.7900000000000p-1, 0x1.7a593f93d7fbcp+0, 0x1.304151f293a81p+0, 0x1.398a131649ea4p+1, 0x1.728d9ed75da1ep+2, 0x1.ee3a07b1736cap+3, 0x1.60106036ec9d4p+5, 0x1.069e8b3e5a09fp+7,
        -0x1.79bbd4e8eb882p-53, 0x1.a7bec94762100p-1, 0x1.7c00000000000p+0, -0x1.a6c06c280445cp-8, 0x1.7b00000000000p-1, 0x1.7cc132eb4e536p+0, 0x1.37bde8bd25d7dp+0, 0x1.456d7a51df797p+1,
        0x1.86858103af33ep+2, 0x1.084f821121c2ep+4, 0x1.7e39a9d7c6de3p+5, 0x1.21664ef4c9a12p+7, 0x1.04d2d39db72ffp-55, 0x1.aab5e13b099b0p-1, 0x1.7c00000000000p+0, 0x1.8265d69ca6c2fp-9,
        0x1.7d00000000000p-1, 0x1.7f386809ba1cdp+0, 0x1.3f83be298b2ebp+0, 0x1.51f62708a6abep+1, 0x1.9be3f090f77abp+2, 0x1.1afe26c13bf38p+4, 0x1.9f7ca65ff02a8p+5, 0x1.3f614da840fe0p+7,
        -0x1.7bde9ab5d1a54p-53, 0x1.adb1d83ebd320p-1, 0x1.8000000000000p+0, -0x1.8f2fec8bc6562p-9, 0x1.7f00000000000p-1, 0x1.81bf7562e1e24p+0, 0x1.4796d469724dbp+0, 0x1.5f2fc86e67917p+1,
        0x1.b2c822f5ae582p+2, 0x1.2f50565ee1919p+4, 0x1.c438f4744d220p+5, 0x1.61003d66309fdp+7, 0x1.470c8fc828894p-55, 0x1.b0b2cd6b287dcp-1, 0x1.8000000000000p+0, 0x1.bf7562e1e23e5p-8,
        0x1.8100000000000p-1, 0x1.8456f9b70ab1dp+0, 0x1.4ffb76d01a674p+0, 0x1.6d27142d7b667p+1, 0x1.cb54905dd4055p+2, 0x1.45723e490ca9bp+4, 0x1.ecd1747c5589bp+5, 0x1.86c463d6db036p+7,
        0x1.4044decf23c2ep+9, -0x1.f09900d173a5fp-56, 0x1.b3b8e10e12d3cp-1, 0x1.8400000000000p+0, 0x1.5be6dc2ac733cp-10, 0x1.8300000000000p-1, 0x1.86ff9cab97b9dp+0, 0x1.58b6404a71b42p+0,
        0x1.7be9e20c0fb6ep+1, 0x1.e5af59b426297p+2, 0x1.5d958013c40eep+4, 0x1.0cea92215e48cp+6, 0x1.b146bb8c0669ap+7, 0x1.68c96fb8eb0fep+9, 0x1.558481fccbad4p-53, 0x1.b6c434bb8ea98p-1,
        0x1.8800000000000p+0, -0x1.0063546846319p-8, 0x1.8500000000000p-1, 0x1.89ba0f71469bfp+0, 0x1.61cc228717efap+0, 0x1.8b874afb7baf7p+1, 0x1.01015ec7286dbp+3, 0x1.77f1f8329a469p+4,
        0x1.25e492927f0ddp+6, 0x1.e135c5ae80cd9p+7, 0x1.9736440df64fdp+9, 0x1.9f53b1ed91b03p-55, 0x1.b9d4eb6067abcp-1, 0x1.8800000000000p+0, 0x1.ba0f71469bf33p-8, 0x1.8700000000000p-1,
        0x1.8c870d797dabfp+0, 0x1.6b426de42d55fp+0, 0x1.9c0fcc0e06552p+1, 0x1.103eceb059907p+3, 0x1.94c6a49a75aa7p+4, 0x1.41a81b2a496d0p+6, 0x1.0baee209cb693p+8, 0x1.cc860285808c5p+9,
        -0x1.e6d8c9b0dc6f3p-53, 0x1.bceb2955ec1c4p-1, 0x1.8c00000000000p+0, 0x1.0e1af2fb57ee7p-9, 0x1.8900000000000p-1, 0x1.8f675d3c502f4p+0, 0x1.751eda3bfb2e4p+0, 0x1.ad956de3987bcp+1,
        0x1.20aa0b30aad0ap+3, 0x1.b45ab16220014p+4, 0x1.60929ec84429cp+6, 0x1.2a5690d747939p+8, 0x1.04f105407f41ep+10, -0x1.75cebfc269962p-57, 0x1.c00714773138cp-1, 0x1.9000000000000p+0,
        -0x1.3145875fa1750p-9, 0x1.8b00000000000p-1, 0x1.925bd111125dfp+0, 0x1.7f6790ad2b4c2p+0, 0x1.c02bf1359a3c8p+1, 0x1.3260188857c21p+3, 0x1.d6feb2515d90ep+4, 0x1.830fad421145ep+6,
        0x1.4d1d6fd789544p+8, 0x1.285614b30ebf1p+10, 0x1.13e7b7876f9d2p-53, 0x1.c328d437f5e74p-1, 0x1.9400000000000p+0, -0x1.a42eeeeda20a4p-8, 0x1.8d00000000000p-1, 0x1.9565481b9477bp+0,
        0x1.8a23367f87779p+0, 0x1.d3e9014665ea0p+1, 0x1.458155a415747p+3, 0x1.fd0e11d7511c0p+4, 0x1.a99b601ec30fbp+6, 0x1.74a72dd7ee7a1p+8, 0x1.514545c2f1724p+10, 0x1.185b3774a5205p-55,
        0x1.c65091bd4ad0cp-1, 0x1.9400000000000p+0, 0x1.65481b9477ac0p-8, 0x1.8f00000000000p-1, 0x1.9884af50630b5p+0, 0x1.9558f94b35a8dp+0, 0x1.e8e46d1a32b1dp+1, 0x1.5a31f0aec68dbp+3,
        0x1.13785fd21a759p+5, 0x1.d4c53f56dfca6p+6, 0x1.a1b45f89c0f5fp+8, 0x1.80bb3c92c8cf3p+10, -0x1.696e8feb6a05ep-55, 0x1.c97e77f82b8ccp-1, 0x1.9800000000000p+0, 0x1.095ea0c6169c6p-9,
        0x1.9100000000000p-1, 0x1.9bbb0292bc29fp+0, 0x1.a1109c8e3d76bp+0, 0x1.ff3868873c480p+1, 0x1.709a6de619c77p+3, 0x1.2a8e95a9417b9p+5, 0x1.0299dbfe20b57p+7, 0x1.d5283e1225431p+8,
        0x1.b7e74c225406cp+10, -0x1.7943174f396dbp-55, 0x1.ccb2b3c239888p-1, 0x1.9c00000000000p+0, -0x1.13f5b50f5839fp-10, 0x1.9300000000000p-1, 0x1.9f094def4783dp+0, 0x1.ad5288e300736p+0,
        0x1.0b80eb2d4d4eep+2, 0x1.88e843f3d0057p+3, 0x1.440d4d20263c0p+5, 0x1.1dd4226e14927p+7, 0x1.0807d5ef13d09p+9, 0x1.f836cfe9e94bep+10, -0x1.13c84e5fd9d2dp-55, 0x1.cfed73fccf104p-1,
        0x1.a000000000000p+0, -0x1.ed642170f854bp-9, 0x1.9500000000000p-1, 0x1.a270aef70c9f9p+0, 0x1.ba27dd12662d9p+0, 0x1.18304e8433b59p+2, 0x1.a34e91b4dd8d9p+3, 0x1.6041f58aa354cp+5,
        0x1.3c82387eb035bp+7, 0x1.29d4e7f89a6b6p+9, 0x1.21b1ab4bed54dp+11, 0x1.55d66fd8283d4p-55, 0x1.d32ee9b2a7684p-1, 0x1.a400000000000p+0, -0x1.8f5108f3606b9p-8, 0x1.9700000000000p-1,
        0x1.a5f2563ea127fp+0, 0x1.c79a81460c218p+0, 0x1.25bc03d14975cp+2, 0x1.c006f2249db66p+3, 0x1.7f856ed0aefcdp+5, 0x1.5f27f2e2028d0p+7, 0x1.50b956ce59595p+9, 0x1.4dc2318c497e2p+11,
        0x1.bdfae76ba54cap-55, 0x1.d677483c60554p-1, 0x1.a400000000000p+0, 0x1.f2563ea127f53p-8, 0x1.9900000000000p-1, 0x1.a98f89061cefep+0, 0x1.d5b53caa1f466p+0, 0x1.34379a92630e8p+2,
        0x1.df52741e37357p+3, 0x1.a23dfd7de2305p+5, 0x1.865fe1911c50fp+7, 0x1.7d981d5ce543dp+9, 0x1.8192e2134a322p+11, -0x1.15cf94fe6dac8p-54, 0x1.d9c6c56821f74p-1, 0x1.a800000000000p+0,
        0x1.8f89061cefdbbp-8, 0x1.9b00000000000p-1, 0x1.ad49a30f0daccp+0, 0x1.e483cddbfee70p+0, 0x1.43b8cc4418459p+2, 0x1.00bd5e6e7e816p+4, 0x1.c8e1a02ee200ep+5, 0x1.b2dfc83038a03p+7,
        0x1.b1814d987e3d9p+9, 0x1.beb1e8827cefap+11, 0x1.8829ae22afce0p-53, 0x1.dd1d99a4c39d0p-1, 0x1.ac00000000000p+0, 0x1.49a30f0dacb86p-8, 0x1.9d00000000000p-1, 0x1.b12218a66e40dp+0,
        0x1.f4130692dc10ap+0, 0x1.5457c64621a80p+2, 0x1.1369aed2a1ab4p+4, 0x1.f3f8dbc003a70p+5, 0x1.e57e1462e99d6p+7, 0x1.edbc2c53f5717p+9, 0x1.0383d0a71e453p+12, 0x1.0af9fbedd86a9p-54,
        0x1.e07c0030cf708p-1, 0x1.b000000000000p+0, 0x1.2218a66e40cbep-8, 0x1.9f00000000000p-1, 0x1.b51a78e9927e5p+0, 0x1.02387581637b3p+1, 0x1.662f7f5b2c17ep+2, 0x1.27ddb36eac07ep+4,
        0x1.12110c70d9c43p+6, 0x1.0f9c488c52943p+8, 0x1.19e9eb1ab4848p+10, 0x1.2e76bb1ec7695p+12, 0x1.a24005e9f6fd9p-53, 0x1.e3e2374dd3c64p-1, 0x1.b400000000000p+0, 0x1.1a78e9927e571p-8,
        0x1.a100000000000p-1, 0x1.b934704e0f95fp+0, 0x1.0ad66ac8dc27bp+1, 0x1.795e1ae05a580p+2, 0x1.3e4fa299aa0a0p+4, 0x1.2d0ada33ab75cp+6, 0x1.309e539d64c89p+8, 0x1.42d39154c34c4p+10,
        0x1.61a5959d15b1dp+12, -0x1.fc899114be565p-53, 0x1.e75080787fd30p-1, 0x1.b800000000000p+0, 0x1.34704e0f95e8bp-8, 0x1.a300000000000p-1, 0x1.bd71cb75f37a1p+0, 0x1.13ebcfc9006e1p+1,
        0x1.8e055c48d2c09p+2, 0x1.56fd7c2c8c9cdp+4, 0x1.4b5576198b971p+6, 0x1.5678c9680f9afp+8, 0x1.72be58af946ddp+10, 0x1.9ede4e1b531f9p+12, -0x1.47f69e4527544p-59, 0x1.eac720a61ad1cp-1,
        0x1.bc00000000000p+0, 0x1.71cb75f37a0dfp-8, 0x1.a500000000000p-1, 0x1.c1d47a5b24f80p+0, 0x1.1d81e7eb9f789p+1, 0x1.a44b2df42b6b7p+2, 0x1.722e5b4766752p+4, 0x1.6d6eeecfadff0p+6,
        0x1.820288b1eb8d5p+8, 0x1.ab0e2ca840144p+10, 0x1.e8614e2126bbfp+12, -0x1.d9a932cc624e2p-55, 0x1.ee466087f8d20p-1, 0x1.c000000000000p+0, 0x1.d47a5b24f8064p-8, 0x1.a700000000000p-1,
        0x1.c65e93de98207p+0, 0x1.27a2e811f641bp+1, 0x1.bc5a3f223266dp+2, 0x1.90340a6ecbe29p+4, 0x1.93eb6c3d499afp+6, 0x1.b43d9ad8cc2f1p+8, 0x1.ed77ca519b816p+10, 0x1.2080a5b3b703bp+13,
        0x1.b187de993c3ddp-56, 0x1.f1ce8cd5a7ce8p-1, 0x1.c800000000000p+0, -0x1.a16c2167df937p-8, 0x1.a900000000000p-1, 0x1.cb1259ca2f05ep+0, 0x1.325a154fc4c95p+1, 0x1.d662bd9c5ff75p+2,
        0x1.b16ce8e93577dp+4, 0x1.bf79ae0e3029ep+6, 0x1.ee61204bcdf91p+8, 0x1.1e0ac31efe3f1p+11, 0x1.5626785df051cp+13, -0x1.d61222d0bc06ep-53, 0x1.f55ff69eab2f0p-1, 0x1.cc00000000000p+0,
        -0x1.db4c6ba1f43e4p-9, 0x1.ab00000000000p-1, 0x1.cff23d56b9f55p+0, 0x1.3db3e86149a3bp+1, 0x1.f29b30b8d0dadp+2, 0x1.d646340e9d1a7p+4, 0x1.f0e89619d6679p+6, 0x1.18f2e92cf3fbcp+9,
        0x1.4cc10844e51bdp+11, 0x1.9762df3a9eb60p+13, 0x1.20e79ef4b1e02p-53, 0x1.f8faf3a4bc01cp-1, 0x1.d000000000000p+0, -0x1.b85528c156248p-13, 0x1.ad00000000000p-1, 0x1.d500e44aad4f2p+0,
        0x1.49be36b85db68p+1, 0x1.08a0be558f351p+3, 0x1.ff3ecc1bcc632p+4, 0x1.149702a555e45p+7, 0x1.404aedd057f33p+9, 0x1.847d922610a18p+11, 0x1.e71463c7aa2b4p+13, -0x1.571d053ca14ecp-54,
        0x1.fc9fdebfaa348p-1, 0x1.d400000000000p+0, 0x1.00e44aad4f267p-8, 0x1.af00000000000p-1, 0x1.da412ec9edc5ap+0, 0x1.5688622b6d908p+1, 0x1.194e0b605b3b4p+3, 0x1.167549338560cp+5,
        0x1.34b7b34b16169p+7, 0x1.6e5083b1baf9cp+9, 0x1.c7475fb9dfbf5p+11, 0x1.2473ef4b4bb01p+14, 0x1.82b31e9f06efcp-53, 0x1.00278c2613f02p+0, 0x1.dc00000000000p+0, -0x1.bed136123a5d1p-8,
        0x1.b100000000000p-1, 0x1.dfb63df3ae0dbp+0, 0x1.6423908ad38cfp+1, 0x1.2b7dbaa166573p+3, 0x1.2ffb438210d3ep+5, 0x1.59862fb634456p+7, 0x1.a45b4ee8f3e34p+9, 0x1.0bd59d39a6c6fp+12,
        0x1.60ccd2b4867e8p+14, -0x1.6097f1cbb85b3p-53, 0x1.020483537e800p+0, 0x1.e000000000000p+0, -0x1.27083147c93edp-10, 0x1.b300000000000p-1, 0x1.e5637b70f5f72p+0, 0x1.72a2eca935102p+1,
        0x1.3f5de43559218p+3, 0x1.4c96eb4e19ca3p+5, 0x1.83d621272dda3p+7, 0x1.e4135c6bfaaedp+9, 0x1.3c717099fb249p+12, 0x1.aba6dd5294f7dp+14, 0x1.488b1c91ffa21p-53, 0x1.03e70b5b309e0p+0,
        0x1.e400000000000p+0, 0x1.637b70f5f723ep-8, 0x1.b500000000000p-1, 0x1.eb4ca21d4b842p+0, 0x1.821bf2be08fc5p+1, 0x1.552386a6a3bd0p+3, 0x1.6cc00bac907e2p+5, 0x1.b4a7894202458p+7,
        0x1.17c35fe065ca6p+10, 0x1.77848e8d5b845p+12, 0x1.048200cd72d76p+15, 0x1.54b6e9cbe508bp-53, 0x1.05cf5e41c2acep+0, 0x1.ec00000000000p+0, -0x1.66bbc568f7c18p-9, 0x1.b700000000000p-1,
        0x1.f175c7fb6eb26p+0, 0x1.92a6ca7ba9c35p+1, 0x1.6d0bc80f5ba9fp+3, 0x1.9104833bd74fbp+5, 0x1.ed31961fce21fp+7, 0x1.44a2e60df5aedp+10, 0x1.bfafc1ac97175p+12, 0x1.3f145c3a8bc22p+15,
        -0x1.94b5da70a42d9p-54, 0x1.07bdb9f358760p+0, 0x1.f000000000000p+0, 0x1.75c7fb6eb2582p-8, 0x1.b900000000000p-1, 0x1.f7e369b29492cp+0, 0x1.a45eb1c35ad8ap+1, 0x1.875d7c8373bb1p+3,
        0x1.ba0d1885e6ae6p+5, 0x1.177840831631ep+8, 0x1.7a4417f51da78p+10, 0x1.0c2b26d7642fbp+13, 0x1.89073594961fbp+15, -0x1.5dece96cdc181p-53, 0x1.09b260a46374ep+0, 0x1.f800000000000p+0,
        -0x1.c964d6b6d3d05p-12, 0x1.bb00000000000p-1, 0x1.fe9a77dd9b1cfp+0, 0x1.b7627b9ae77afp+1, 0x1.a46b03338306dp+3, 0x1.e8a38a0caace9p+5, 0x1.3ddbb864f53a2p+8, 0x1.baaf0d6c97f8dp+10,
        0x1.42eefdfae5a98p+13, 0x1.e701ae19501dap+15, 0x1.cc4f4c7d3d675p-54, 0x1.0bad993ec49aep+0, 0x1.0000000000000p+1, -0x1.65882264e310dp-8, 0x1.bd00000000000p-1, 0x1.02d0334302f3bp+1,
        0x1.cbd527f5aaf0dp+1, 0x1.c49490c635c0ap+3, 0x1.0edd1b6bb1732p+6, 0x1.6ae3d9691a9f4p+8, 0x1.043c761482fc6p+11, 0x1.87037f81eb6e0p+13, 0x1.2fa30e84fe55ep+16, -0x1.820f1228fc41dp-54,
        0x1.0dafafdd4ae68p+0, 0x1.0200000000000p+1, 0x1.a0668605e76b0p-8, 0x1.bf00000000000p-1, 0x1.067d9f9c947a3p+1, 0x1.e1de9a1722882p+1, 0x1.e84b041fe0247p+3, 0x1.2d3aedbd1d676p+6,
        0x1.9ff78e088bef5p+8, 0x1.3378064d9a484p+11, 0x1.dc32d1974f9b5p+13, 0x1.7d295ce268611p+16, -0x1.5a192d437d23fp-53, 0x1.0fb8f657efdcap+0, 0x1.0600000000000p+1, 0x1.f67e7251e8cf3p-9,
        0x1.c100000000000p-1, 0x1.0a58db1fffa6dp+1, 0x1.f9ac74e7307c3p+1, 0x1.0809b5ea15962p+4, 0x1.501d05418e1b6p+6, 0x1.ded80b476d79fp+8, 0x1.6d2bf37f33d5fp+11, 0x1.23c31a43f6c6fp+14,
        0x1.e1e46db17bbaap+16, -0x1.7eb6241d8ad56p-53, 0x1.11c9c4e3ade0ap+0, 0x1.0a00000000000p+1, 0x1.636c7ffe9b457p-9, 0x1.c300000000000p-1, 0x1.0e65a1d1bdcc6p+1, 0x1.09b993503cccep+2,
        0x1.1e45b7580ec24p+4, 0x1.785ca1803e176p+6, 0x1.14ddb8458a77dp+9, 0x1.b41d96c115ab7p+11, 0x1.67df0d7bce584p+14, 0x1.32ef5f5487646p+17, -0x1.c4040f3631254p-54, 0x1.13e27ac964da8p+0,
        0x1.0e00000000000p+1, 0x1.968746f731770p-9, 0x1.c500000000000p-1, 0x1.12a82068fbcb4p+1, 0x1.17b797fe89a5fp+2, 0x1.37376d37f3897p+4, 0x1.a704edf3b47a2p+6, 0x1.41b83eb114449p+9,
        0x1.05f758d323120p+12, 0x1.befec8ae65dddp+14, 0x1.8a2a2d1814341p+17, 0x1.3e83dfb25ec76p-53, 0x1.16037f37ffedap+0, 0x1.1200000000000p+1, 0x1.5040d1f796787p-8, 0x1.c700000000000p-1,
        0x1.172505f8f574bp+1, 0x1.26f35b566493dp+2, 0x1.534f595186e3dp+4, 0x1.dd60b947d5ea5p+6, 0x1.77c77568c5d73p+9, 0x1.3cb66a26261f0p+12, 0x1.17b06bf32194dp+15, 0x1.fe92111490e42p+17,
        -0x1.344285376cb61p-53, 0x1.182d4236fe314p+0, 0x1.1800000000000p+1, -0x1.b5f40e15169a9p-8, 0x1.c900000000000p-1, 0x1.1be1991b4c8d8p+1, 0x1.3795bbe69bae6p+2, 0x1.73151cd6f8b02p+4,
        0x1.0e864d86a7bffp+7, 0x1.b95cc515f5bd6p+9, 0x1.8180cd070b4a1p+12, 0x1.60d2cc9b24d80p+15, 0x1.4dbf6aa392cafp+18, -0x1.89bd0f5844c55p-53, 0x1.1a603dbfaf236p+0, 0x1.1c00000000000p+1,
        -0x1.e66e4b37285fcp-11, 0x1.cb00000000000p-1, 0x1.20e3d1757f6b1p+1, 0x1.49ce9ae890640p+2, 0x1.972d4d6174f60p+4, 0x1.340798c82df92p+7, 0x1.04be6acab5569p+10, 0x1.d8a99b362e75ap+12,
        0x1.c0ed7389374dcp+15, 0x1.b8adfca5e9653p+18, -0x1.0cbc74a1e3e49p-55, 0x1.1c9cf704f5d26p+0, 0x1.2000000000000p+1, 0x1.c7a2eafed62a2p-8, 0x1.cd00000000000p-1, 0x1.263276b3395aap+1,
        0x1.5dd6633fb1467p+2, 0x1.c0610dcf3437cp+4, 0x1.607cec9d7c47ap+7, 0x1.360fba330dc5cp+10, 0x1.240b438a3194bp+13, 0x1.20437baa6a879p+16, 0x1.2610604d6f19cp+19, 0x1.bccf515e5252cp-53,
        0x1.1ee3fff35681ap+0, 0x1.2600000000000p+1, 0x1.93b599cad4ce9p-10, 0x1.cf00000000000p-1, 0x1.2bd54664a8350p+1, 0x1.73eff945190a0p+2, 0x1.efa80c7cc5224p+4, 0x1.958aa896f1658p+7,
        0x1.734504fd54e04p+10, 0x1.6bf554cd60c4ap+13, 0x1.75ebe3effd07cp+16, 0x1.8d03c9e2e6981p+19, -0x1.87ceec8a488ffp-54, 0x1.2135f8f597306p+0, 0x1.2c00000000000p+1, -0x1.55ccdabe583fep-10,
        0x1.d100000000000p-1, 0x1.31d522a40ea5cp+1, 0x1.8c6b352b4947dp+2, 0x1.131ae5d01146ep+5, 0x1.d54fb0163e71cp+7, 0x1.bfe8aef3ed15bp+10, 0x1.c9c28a33a6b00p+13, 0x1.ea5231456e1a6p+16,
        0x1.0f60ffc8790dbp+20, 0x1.c104f6fabca41p-53, 0x1.2393930d87c68p+0, 0x1.3200000000000p+1, -0x1.56eadf8ad1cf9p-10, 0x1.d300000000000p-1, 0x1.383c4c053c623p+1, 0x1.a7a816adbff2cp+2,
        0x1.32c5be219a24ep+5, 0x1.1148430f4b8d8p+8, 0x1.10659bc59423ep+11, 0x1.22c093d537ae5p+14, 0x1.454a2a4b7d930p+17, 0x1.78151c151f3c3p+20, -0x1.2f226779e9951p-53, 0x1.25fd9254e3f9cp+0,
        0x1.3800000000000p+1, 0x1.e26029e311a8bp-10, 0x1.d500000000000p-1, 0x1.3f16aa2f65f8cp+1, 0x1.c61af36c0308ep+2, 0x1.57c825337ff7dp+5, 0x1.407a37fb84ba9p+8, 0x1.4e4764c74dea7p+11,
        0x1.75638df1c2124p+14, 0x1.b5320a2556e94p+17, 0x1.087cd7d68abbep+21, -0x1.cd58c73a87ab9p-53, 0x1.2874d10017b06p+0, 0x1.4000000000000p+1, -0x1.d2aba1340e849p-8, 0x1.d700000000000p-1,
        0x1.467227ba9a810p+1, 0x1.e851fcbc74735p+2, 0x1.83596f3879985p+5, 0x1.7aaebcd297f00p+8, 0x1.9e3c831669f50p+11, 0x1.e5420b7cbb664p+14, 0x1.29f7bb75100a0p+18, 0x1.7a1c151d127bfp+21,
        0x1.c647e46d9c78fp-53, 0x1.2afa4304962aep+0, 0x1.4600000000000p+1, 0x1.c89eea6a041c9p-9, 0x1.d900000000000p-1, 0x1.4e5f27a99a835p+1, 0x1.077e515b0232dp+3, 0x1.b70c1ee468866p+5,
        0x1.c334a43a041c3p+8, 0x1.036d153d2c164p+12, 0x1.3f7b110ccedbep+15, 0x1.9c160f6c2e560p+18, 0x1.131dd6d21d20fp+22, 0x1.786832ec50766p+25, -0x1.95596d1134eccp-53, 0x1.2d8efa8f4b028p+0,
        0x1.4e00000000000p+1, 0x1.7c9ea66a0d2c7p-9, 0x1.db00000000000p-1, 0x1.56f11d6373b90p+1, 0x1.1d7acc3747df3p+3, 0x1.f4ef36a014d6fp+5, 0x1.0f4c4505c454bp+9, 0x1.48d16214975c5p+12,
        0x1.aacfdf57bfac6p+15, 0x1.222355225a6edp+19, 0x1.98643acba67abp+22, 0x1.267b9de5d19b9p+26, -0x1.ef63c42c92439p-53, 0x1.30342d86bed76p+0, 0x1.5600000000000p+1, 0x1.e23ac6e771f48p-8,
        0x1.dd00000000000p-1, 0x1.603f53d2d8cf1p+1, 0x1.36a84ef4a10fap+3, 0x1.1fdf34ea265afp+6, 0x1.499b5d944f636p+9, 0x1.a64b837f73bacp+12, 0x1.21b9f259b27fcp+16, 0x1.a0669265d5b9fp+19,
        0x1.35d8e3dc806e2p+23, 0x1.d865736ad8b00p+26, 0x1.4ceeb3ffcdca3p-53, 0x1.32eb3c69d2d10p+0, 0x1.6000000000000p+1, 0x1.fa9e96c678625p-10, 0x1.df00000000000p-1, 0x1.6a65f5fcdf915p+1,
        0x1.53b6e68321bdap+3, 0x1.4d949706e8da9p+6, 0x1.950ef4a70d2d7p+9, 0x1.133191f15e14ep+13, 0x1.907b1846a9bd5p+16, 0x1.3139c17c39016p+20, 0x1.e1da3bc86f11bp+23, 0x1.8597fd9f86f3bp+27,
        0x1.2d4f87d0d5190p-60, 0x1.35b5bafa88354p+0, 0x1.6a00000000000p+1, 0x1.97d7f37e455fdp-9, 0x1.e100000000000p-1, 0x1.7587741d1dbf9p+1, 0x1.758a8f5852184p+3, 0x1.861ee65c0f467p+6,
        0x1.f83c0d2d91276p+9, 0x1.6ca7c43ec3b0ep+13, 0x1.1a722718322c8p+17, 0x1.ca4c69533d806p+20, 0x1.812b7e9899583p+24, 0x1.4b87585ee8b86p+28, -0x1.9defbd1aeeed1p-54, 0x1.38957b510476ep+0,
        0x1.7600000000000p+1, -0x1.e22f8b8901bf9p-9, 0x1.e300000000000p-1, 0x1.81ce6e1c37e57p+1, 0x1.9d4f3d3dc9910p+3, 0x1.cd074e3095065p+6, 0x1.3e764c5c38224p+10, 0x1.ec5ae3cae1f31p+13,
        0x1.97a50c0645f38p+17, 0x1.61866d8a7f25ep+21, 0x1.3daf58c2f04a3p+25, 0x1.2450ea9143c1fp+29, 0x1.d25be9fd995bcp-56, 0x1.3b8c9c35d33e6p+0, 0x1.8200000000000p+1, -0x1.8c8f1e40d49e0p-10,
        0x1.e500000000000p-1, 0x1.8f706285640bbp+1, 0x1.cc96b3b2b7cd1p+3, 0x1.13adfc5341328p+7, 0x1.9908d16e928a9p+10, 0x1.539867cc08a3cp+14, 0x1.2dfc531dd3e45p+18, 0x1.19499e2a13787p+22,
        0x1.0f943f94424adp+26, 0x1.0c6bccdcd49bep+30, -0x1.e24586d41701dp-54, 0x1.3e9d9c088bd28p+0, 0x1.9000000000000p+1, -0x1.1f3af537e8a00p-8, 0x1.e700000000000p-1, 0x1.9eb186562d1e0p+1,
        0x1.02c3175651223p+4, 0x1.4e431336e41c7p+7, 0x1.0bca6a065da69p+11, 0x1.e034d917af357p+14, 0x1.cd2c14168fb0fp+18, 0x1.cfeb615bb794dp+22, 0x1.e3ee16effd5e5p+26, 0x1.024e71acb4d9cp+31,
        -0x1.c29c8d93f153fp-54, 0x1.41cb72183e810p+0, 0x1.9e00000000000p+1, 0x1.630cac5a3c038p-8, 0x1.e900000000000p-1, 0x1.afea6a364196fp+1, 0x1.258f30b19a2ebp+4, 0x1.9bda52520ac75p+7,
        0x1.669bc8f67edeap+11, 0x1.5d78cc026c9f8p+15, 0x1.6ccb41e3b36c2p+19, 0x1.8ede4bf45c805p+23, 0x1.c2f6a8ac89e76p+27, 0x1.0675e4ca9eb55p+32, 0x1.36ac10d13e3dfp+36, 0x1.b1d74f2de93a6p-54,
        0x1.4519b155fb22ep+0, 0x1.b000000000000p+1, -0x1.595c9be690e67p-11, 0x1.eb00000000000p-1, 0x1.c39084bd1c065p+1, 0x1.50d8826c39ffdp+4, 0x1.0296b69d3e79ep+8, 0x1.ed279d7feea5dp+11,
        0x1.072a8fd5bd547p+16, 0x1.2cdb94a08bb38p+20, 0x1.68482536bed06p+24, 0x1.be1ff2f10e88dp+28, 0x1.1c966abdbbdacp+33, 0x1.7101102e62ddap+37, -0x1.0855d3e907e71p-53, 0x1.488cb8fa73920p+0,
        0x1.c400000000000p+1, -0x1.bded0b8fe6ddfp-9, 0x1.ed00000000000p-1, 0x1.da43912aaf9a9p+1, 0x1.87d4662f25109p+4, 0x1.4c3393f133a3fp+8, 0x1.5e143662036f9p+12, 0x1.9cf0474467831p+16,
        0x1.04e10576c6fa8p+21, 0x1.59489ff4f8e88p+25, 0x1.d88d2b44962a9p+29, 0x1.4d83897a288f3p+34, 0x1.de10b6cf738b3p+38, -0x1.e9ea75f7263ccp-55, 0x1.4c29faa786f36p+0, 0x1.da00000000000p+1,
        0x1.0e44aabe6a2adp-9, 0x1.ef00000000000p-1, 0x1.f4e35c169b52fp+1, 0x1.cf77329e8699cp+4, 0x1.b6d37fc1818d6p+8, 0x1.026551386790ap+13, 0x1.54a1f4ff79d1ep+17, 0x1.e104a7db0265ap+21,
        0x1.63c39e5c8114bp+26, 0x1.10156f52a87dbp+31, 0x1.add762e9e7abep+35, 0x1.586ab6ec81361p+40, 0x1.35690e395eea6p-54, 0x1.4ff862e5965a2p+0, 0x1.f400000000000p+1, 0x1.c6b82d36a5e70p-8,
    };
    return tab;
}
RDR_FN const double *inroot_table() {
    static const double tab[128] = {
        0x1.68a1f80d71820p+0, 0x1.65de82af9631fp+0, 0x1.632b1201d39e5p+0, 0x1.60870d91bf3c2p+0, 0x1.5df1e4be5e7a2p+0, 0x1.5b6b0e361668bp+0, 0x1.58f2077eca742p+0, 0x1.568654873c1e3p+0,
        0x1.54277f40d6cb8p+0, 0x1.51d51741283a0p+0, 0x1.4f8eb16a59835p+0, 0x1.4d53e79a0e15cp+0, 0x1.4b24585e1ca16p+0, 0x1.48ffa6aea45fep+0, 0x1.46e579ad0c47ap+0, 0x1.44d57c6785455p+0,
        0x1.42cf5da0b1da2p+0, 0x1.40d2cf9b1e1cbp+0, 0x1.3edf87e83b18ep+0, 0x1.3cf53f3a97317p+0, 0x1.3b13b13b13b10p+0, 0x1.393a9c60dd0e1p+0, 0x1.3769c1cbf0b05p+0, 0x1.35a0e521ff992p+0,
        0x1.33dfcc6d81367p+0, 0x1.32263ffecdbddp+0, 0x1.30740a4f1a8e0p+0, 0x1.2ec8f7e53640dp+0, 0x1.2d24d73be4ec9p+0, 0x1.2b8778a9bf8ecp+0, 0x1.29f0ae4a7bd6fp+0, 0x1.28604be983e91p+0,
        0x1.26d626edc7332p+0, 0x1.25521646af854p+0, 0x1.23d3f25a271bcp+0, 0x1.225b94f39da5bp+0, 0x1.20e8d933fbc46p+0, 0x1.1f7b9b8275ccep+0, 0x1.1e13b97e2f7f2p+0, 0x1.1cb111f0a37d1p+0,
        0x1.1b5384c0c28bcp+0, 0x1.19faf2e6beff6p+0, 0x1.18a73e6079ee1p+0, 0x1.17584a268851ep+0, 0x1.160dfa21c700dp+0, 0x1.14c8332174f8cp+0, 0x1.1386dad1cc065p+0, 0x1.1249d7b3107fbp+0,
        0x1.1111111111120p+0, 0x1.0fdc6efb1043ep+0, 0x1.0eabda3c11a68p+0, 0x1.0d7f3c53851c1p+0, 0x1.0c567f6e4acefp+0, 0x1.0b318e600b285p+0, 0x1.0a10549cddf4fp+0, 0x1.08f2be333c85cp+0,
        0x1.07d8b7c63aadfp+0, 0x1.06c22e8802d59p+0, 0x1.05af103491a22p+0, 0x1.049f4b0cadb27p+0, 0x1.0392cdd118774p+0, 0x1.028987bdf5125p+0, 0x1.0183688662733p+0, 0x1.00806050463f7p+0,
        0x1.fe02fb08b05a2p-1, 0x1.fa1a7bb61d36fp-1, 0x1.f648a3a321fb7p-1, 0x1.f28c9b380eba0p-1, 0x1.eee595eba94c0p-1, 0x1.eb52d18b3d37dp-1, 0x1.e7d3959112457p-1, 0x1.e4673287f9dc5p-1,
        0x1.e10d017ac51a6p-1, 0x1.ddc4636e9573ep-1, 0x1.da8cc0e7149eep-1, 0x1.d7658973b887cp-1, 0x1.d44e33454e185p-1, 0x1.d1463acb186adp-1, 0x1.ce4d2256e352ep-1, 0x1.cb6271c77707cp-1,
        0x1.c885b638e9087p-1, 0x1.c5b681ba51d60p-1, 0x1.c2f46b087a5e7p-1, 0x1.c03f0d4d1e1b0p-1, 0x1.bd9607e2670d9p-1, 0x1.baf8fe1a51499p-1, 0x1.b8679709aaac2p-1, 0x1.b5e17d566a055p-1,
        0x1.b3665f091e62bp-1, 0x1.b0f5ed613d3f1p-1, 0x1.ae8fdcac1a25dp-1, 0x1.ac33e41e57a2dp-1, 0x1.a9e1bdafa4be1p-1, 0x1.a79925f89def6p-1, 0x1.a559dc12abd3ap-1, 0x1.a323a179bcde9p-1,
        0x1.a0f639efb9e4bp-1, 0x1.9ed16b6197fd7p-1, 0x1.9cb4fdcdec59dp-1, 0x1.9aa0bb2ce89c8p-1, 0x1.98946f59a8bebp-1, 0x1.968fe7fcbc524p-1, 0x1.9492f477d78d4p-1, 0x1.929d65d2990a8p-1,
        0x1.90af0ea853815p-1, 0x1.8ec7c316cae0fp-1, 0x1.8ce758add6570p-1, 0x1.8b0da65fd9394p-1, 0x1.893a847305ac9p-1, 0x1.876dcc735d942p-1, 0x1.85a75925660f9p-1, 0x1.83e706798349dp-1,
        0x1.822cb17ff2eb2p-1, 0x1.8078385d5c0dep-1, 0x1.7ec97a3fec010p-1, 0x1.7d205754f8264p-1, 0x1.7b7cb0bf1d70fp-1, 0x1.79de688cd6672p-1, 0x1.784561af81485p-1, 0x1.76b17ff2d032cp-1,
        0x1.7522a7f49d736p-1, 0x1.7398bf1d1ee54p-1, 0x1.7213ab9772ee3p-1, 0x1.7093544a82bc7p-1, 0x1.6f17a0d23510ep-1, 0x1.6da07978eda76p-1, 0x1.6c2dc73154e64p-1, 0x1.6abf7390648fep-1,
    };
    return tab;
}

namespace detail {
// e_asin.c: the seven intervals of [0.125, 0.96875) share one shape: a series in xx = |x| - point with NC + 1 coefficients
// a[1] ... a[NC + 1], the tail and the head of asin(point) behind them
template <int NC>
RDR_FN double acos_interval(const double *a, double x, bool pos) {
    const double xx = (pos ? x : -x) - a[0];
    double q = a[NC + 1];
    for (int j = NC; j >= 2; --j) q = fma(xx, q, a[j]);
    const double t = fma(xx, a[1], fma(xx * xx, q, a[NC + 2]));
    return pos ? ((kHpi1 - t) + (kHpi - a[NC + 3])) : ((t + kHpi1) + (a[NC + 3] + kHpi));
}
} // namespace detail

// e_asin.c: __ieee754_acos
RDR_FN double acos(double x) {
    using namespace detail;
    const int32_t m = (int32_t)hi_word(x);
    const int32_t k = m & 0x7fffffff;
    const double f1 = 0x1.55555555554f9p-3, f2 = 0x1.333333336127dp-4, f3 = 0x1.6db6dae42c0e4p-5, f4 = 0x1.f1c7e04f4ad99p-6,
                 f5 = 0x1.6e442c822d419p-6, f6 = 0x1.292d80f453c72p-6;
    if (k < 0x3c880000) return kHpi;
    if (k < 0x3fc00000) {
        const double x2 = x * x;
        const double p = fma(x2, fma(x2, fma(x2, fma(x2, fma(x2, f6, f5), f4), f3), f2), f1);
        const double r = kHpi - x;
        const double cor = fma(-p, x * x2, ((kHpi - r) - x) + kHpi1);
        return r + cor;
    }
    const double *a = asin_table();
    const bool pos = m > 0;
    if (k < 0x3fd00000) return acos_interval<5>(a + 11 * ((k >> 15) & 0x1f), x, pos);
    if (k < 0x3fe00000) return acos_interval<5>(a + 352 + 11 * ((k >> 14) & 0x3f), x, pos);
    if (k < 0x3fe80000) return acos_interval<6>(a + 1056 + 12 * ((k >> 13) & 0x7f), x, pos);
    if (k < 0x3fed8000) return acos_interval<7>(a + 992 + 13 * ((k >> 13) & 0x7f), x, pos);
    if (k < 0x3fee8000) return acos_interval<8>(a + 884 + 14 * ((k >> 13) & 0x7f), x, pos);
    if (k < 0x3fef0000) return acos_interval<9>(a + 768 + 15 * ((k >> 13) & 0x7f), x, pos);
    if (k < 0x3ff00000) {                                   // 0.96875 <= |x| < 1: through sqrt((1 - |x|) / 2)
        const double z = 0.5 * (pos ? (1.0 - x) : (1.0 + x));
        const int32_t kz = (int32_t)hi_word(z);
        double t = inroot_table()[(kz & 0x001fffff) >> 14] * bits2d((uint64_t)(1023 + 511 - (kz >> 21)) << 52);
        const double r = fma(-(t * t), z, 1.0);
        t = fma(r, fma(r, fma(r, 0x1.4006318d1dab9p-2, 0x1.800496769c91ap-2), 0x1.fffffff757304p-2), 0x1.fffffffecc1ddp-1) * t;
        const double c = z * t;
        const double w = fma(-(t * 0.5), c, 1.5);
        const double t27 = 0x1p+27;
        const double y = fma(-t27, c, fma(c, t27, c));
        const double cc = fma(-y, y, z) / fma(w, c, y);
        const double p = fma(z, fma(z, fma(z, fma(z, fma(z, f6, f5), f4), f3), f2), f1) * z;
        const double pc = p * (y + cc);
        if (m < 0) {
            const double res = ((kHpi1 - cc) - pc) + (kHpi - y);
            return res + res;
        }
        const double res = (cc + pc) + y;
        return res + res;
    }
    if (k == 0x3ff00000 && lo_word(x) == 0) return pos ? 0.0 : kOpi;
    if (k > 0x7ff00000 || (k == 0x7ff00000 && lo_word(x) != 0)) return x + x;
    return (x - x) / (x - x);                               // |x| > 1 (finite or infinite): invalid
}

// {1 / c, log(c)} for the 128 intervals of [0x1.6p-1, 0x1.6p0) -- glibc's __log_data.tab (e_log_data.c)
RDR_FN const double *log_table() {
    static const double tab[256] = {
        0x1.734f0c3e0de9fp+0, -0x1.7cc7f79e69000p-2, 0x1.713786a2ce91fp+0, -0x1.76feec20d0000p-2,
        0x1.6f26008fab5a0p+0, -0x1.713e31351e000p-2, 0x1.6d1a61f138c7dp+0, -0x1.6b85b38287800p-2,
        0x1.6b1490bc5b4d1p+0, -0x1.65d5590807800p-2, 0x1.69147332f0cbap+0, -0x1.602d076180000p-2,
        0x1.6719f18224223p+0, -0x1.5a8ca86909000p-2, 0x1.6524f99a51ed9p+0, -0x1.54f4356035000p-2,
        0x1.63356aa8f24c4p+0, -0x1.4f637c36b4000p-2, 0x1.614b36b9ddc14p+0, -0x1.49da7fda85000p-2,
        0x1.5f66452c65c4cp+0, -0x1.445923989a800p-2, 0x1.5d867b5912c4fp+0, -0x1.3edf439b0b800p-2,
        0x1.5babccb5b90dep+0, -0x1.396ce448f7000p-2, 0x1.59d61f2d91a78p+0, -0x1.3401e17bda000p-2,
        0x1.5805612465687p+0, -0x1.2e9e2ef468000p-2, 0x1.56397cee76bd3p+0, -0x1.2941b3830e000p-2,
        0x1.54725e2a77f93p+0, -0x1.23ec58cda8800p-2, 0x1.52aff42064583p+0, -0x1.1e9e129279000p-2,
        0x1.50f22dbb2bddfp+0, -0x1.1956d2b48f800p-2, 0x1.4f38f4734ded7p+0, -0x1.141679ab9f800p-2,
        0x1.4d843cfde2840p+0, -0x1.0edd094ef9800p-2, 0x1.4bd3ec078a3c8p+0, -0x1.09aa518db1000p-2,
        0x1.4a27fc3e0258ap+0, -0x1.047e65263b800p-2, 0x1.4880524d48434p+0, -0x1.feb224586f000p-3,
        0x1.46dce1b192d0bp+0, -0x1.f474a7517b000p-3, 0x1.453d9d3391854p+0, -0x1.ea4443d103000p-3,
        0x1.43a2744b4845ap+0, -0x1.e020d44e9b000p-3, 0x1.420b54115f8fbp+0, -0x1.d60a22977f000p-3,
        0x1.40782da3ef4b1p+0, -0x1.cc00104959000p-3, 0x1.3ee8f5d57fe8fp+0, -0x1.c202956891000p-3,
        0x1.3d5d9a00b4ce9p+0, -0x1.b81178d811000p-3, 0x1.3bd60c010c12bp+0, -0x1.ae2c9ccd3d000p-3,
        0x1.3a5242b75dab8p+0, -0x1.a45402e129000p-3, 0x1.38d22cd9fd002p+0, -0x1.9a877681df000p-3,
        0x1.3755bc5847a1cp+0, -0x1.90c6d69483000p-3, 0x1.35dce49ad36e2p+0, -0x1.87120a645c000p-3,
        0x1.34679984dd440p+0, -0x1.7d68fb4143000p-3, 0x1.32f5cceffcb24p+0, -0x1.73cb83c627000p-3,
        0x1.3187775a10d49p+0, -0x1.6a39a9b376000p-3, 0x1.301c8373e3990p+0, -0x1.60b3154b7a000p-3,
        0x1.2eb4ebb95f841p+0, -0x1.5737d76243000p-3, 0x1.2d50a0219a9d1p+0, -0x1.4dc7b8fc23000p-3,
        0x1.2bef9a8b7fd2ap+0, -0x1.4462c51d20000p-3, 0x1.2a91c7a0c1babp+0, -0x1.3b08abc830000p-3,
        0x1.293726014b530p+0, -0x1.31b996b490000p-3, 0x1.27dfa5757a1f5p+0, -0x1.2875490a44000p-3,
        0x1.268b39b1d3bbfp+0, -0x1.1f3b9f879a000p-3, 0x1.2539d838ff5bdp+0, -0x1.160c8252ca000p-3,
        0x1.23eb7aac9083bp+0, -0x1.0ce7f57f72000p-3, 0x1.22a012ba940b6p+0, -0x1.03cdc49fea000p-3,
        0x1.2157996cc4132p+0, -0x1.f57bdbc4b8000p-4, 0x1.201201dd2fc9bp+0, -0x1.e370896404000p-4,
        0x1.1ecf4494d480bp+0, -0x1.d17983ef94000p-4, 0x1.1d8f5528f6569p+0, -0x1.bf9674ed8a000p-4,
        0x1.1c52311577e7cp+0, -0x1.adc79202f6000p-4, 0x1.1b17c74cb26e9p+0, -0x1.9c0c3e7288000p-4,
        0x1.19e010c2c1ab6p+0, -0x1.8a646b372c000p-4, 0x1.18ab07bb670bdp+0, -0x1.78d01b3ac0000p-4,
        0x1.1778a25efbcb6p+0, -0x1.674f145380000p-4, 0x1.1648d354c31dap+0, -0x1.55e0e6d878000p-4,
        0x1.151b990275fddp+0, -0x1.4485cdea1e000p-4, 0x1.13f0ea432d24cp+0, -0x1.333d94d6aa000p-4,
        0x1.12c8b7210f9dap+0, -0x1.22079f8c56000p-4, 0x1.11a3028ecb531p+0, -0x1.10e4698622000p-4,
        0x1.107fbda8434afp+0, -0x1.ffa6c6ad20000p-5, 0x1.0f5ee0f4e6bb3p+0, -0x1.dda8d4a774000p-5,
        0x1.0e4065d2a9fcep+0, -0x1.bbcece4850000p-5, 0x1.0d244632ca521p+0, -0x1.9a1894012c000p-5,
        0x1.0c0a77ce2981ap+0, -0x1.788583302c000p-5, 0x1.0af2f83c636d1p+0, -0x1.5715e67d68000p-5,
        0x1.09ddb98a01339p+0, -0x1.35c8a49658000p-5, 0x1.08cabaf52e7dfp+0, -0x1.149e364154000p-5,
        0x1.07b9f2f4e28fbp+0, -0x1.e72c082eb8000p-6, 0x1.06ab58c358f19p+0, -0x1.a55f152528000p-6,
        0x1.059eea5ecf92cp+0, -0x1.63d62cf818000p-6, 0x1.04949cdd12c90p+0, -0x1.228fb8caa0000p-6,
        0x1.038c6c6f0ada9p+0, -0x1.c317b20f90000p-7, 0x1.02865137932a9p+0, -0x1.419355daa0000p-7,
        0x1.0182427ea7348p+0, -0x1.81203c2ec0000p-8, 0x1.008040614b195p+0, -0x1.0040979240000p-9,
        0x1.fe01ff726fa1ap-1, 0x1.feff384900000p-9, 0x1.fa11cc261ea74p-1, 0x1.7dc41353d0000p-7,
        0x1.f6310b081992ep-1, 0x1.3cea3c4c28000p-6, 0x1.f25f63ceeadcdp-1, 0x1.b9fc114890000p-6,
        0x1.ee9c8039113e7p-1, 0x1.1b0d8ce110000p-5, 0x1.eae8078cbb1abp-1, 0x1.58a5bd001c000p-5,
        0x1.e741aa29d0c9bp-1, 0x1.95c8340d88000p-5, 0x1.e3a91830a99b5p-1, 0x1.d276aef578000p-5,
        0x1.e01e009609a56p-1, 0x1.07598e598c000p-4, 0x1.dca01e577bb98p-1, 0x1.253f5e30d2000p-4,
        0x1.d92f20b7c9103p-1, 0x1.42edd8b380000p-4, 0x1.d5cac66fb5ccep-1, 0x1.606598757c000p-4,
        0x1.d272caa5ede9dp-1, 0x1.7da76356a0000p-4, 0x1.cf26e3e6b2ccdp-1, 0x1.9ab434e1c6000p-4,
        0x1.cbe6da2a77902p-1, 0x1.b78c7bb0d6000p-4, 0x1.c8b266d37086dp-1, 0x1.d431332e72000p-4,
        0x1.c5894bd5d5804p-1, 0x1.f0a3171de6000p-4, 0x1.c26b533bb9f8cp-1, 0x1.067152b914000p-3,
        0x1.bf583eeece73fp-1, 0x1.147858292b000p-3, 0x1.bc4fd75db96c1p-1, 0x1.2266ecdca3000p-3,
        0x1.b951e0c864a28p-1, 0x1.303d7a6c55000p-3, 0x1.b65e2c5ef3e2cp-1, 0x1.3dfc33c331000p-3,
        0x1.b374867c9888bp-1, 0x1.4ba366b7a8000p-3, 0x1.b094b211d304ap-1, 0x1.5933928d1f000p-3,
        0x1.adbe885f2ef7ep-1, 0x1.66acd2418f000p-3, 0x1.aaf1d31603da2p-1, 0x1.740f8ec669000p-3,
        0x1.a82e63fd358a7p-1, 0x1.815c0f51af000p-3, 0x1.a5740ef09738bp-1, 0x1.8e92954f68000p-3,
        0x1.a2c2a90ab4b27p-1, 0x1.9bb3602f84000p-3, 0x1.a01a01393f2d1p-1, 0x1.a8bed1c2c0000p-3,
        0x1.9d79f24db3c1bp-1, 0x1.b5b515c01d000p-3, 0x1.9ae2505c7b190p-1, 0x1.c2967ccbcc000p-3,
        0x1.9852ef297ce2fp-1, 0x1.cf635d5486000p-3, 0x1.95cbaeea44b75p-1, 0x1.dc1bd3446c000p-3,
        0x1.934c69de74838p-1, 0x1.e8c01b8cfe000p-3, 0x1.90d4f2f6752e6p-1, 0x1.f5509c0179000p-3,
        0x1.8e6528effd79dp-1, 0x1.00e6c121fb800p-2, 0x1.8bfce9fcc007cp-1, 0x1.071b80e93d000p-2,
        0x1.899c0dabec30ep-1, 0x1.0d46b9e867000p-2, 0x1.87427aa2317fbp-1, 0x1.13687334bd000p-2,
        0x1.84f00acb39a08p-1, 0x1.1980d67234800p-2, 0x1.82a49e8653e55p-1, 0x1.1f8ffe0cc8000p-2,
        0x1.8060195f40260p-1, 0x1.2595fd7636800p-2, 0x1.7e22563e0a329p-1, 0x1.2b9300914a800p-2,
        0x1.7beb377dcb5adp-1, 0x1.3187210436000p-2, 0x1.79baa679725c2p-1, 0x1.377266dec1800p-2,
        0x1.77907f2170657p-1, 0x1.3d54ffbaf3000p-2, 0x1.756cadbd6130cp-1, 0x1.432eee32fe000p-2,
    };
    return tab;
}
// {1 / c, 0, log(c), tail of log(c)} -- __pow_log_data.tab (e_pow_log_data.c)
RDR_FN const double *pow_log_table() {
    static const double tab[512] = {
        0x1.6a00000000000p+0, 0x0.0p+0, -0x1.62c82f2b9c800p-2, 0x1.ab42428375680p-48,
        0x1.6800000000000p+0, 0x0.0p+0, -0x1.5d1bdbf580800p-2, -0x1.ca508d8e0f720p-46,
        0x1.6600000000000p+0, 0x0.0p+0, -0x1.5767717455800p-2, -0x1.362a4d5b6506dp-45,
        0x1.6400000000000p+0, 0x0.0p+0, -0x1.51aad872df800p-2, -0x1.684e49eb067d5p-49,
        0x1.6200000000000p+0, 0x0.0p+0, -0x1.4be5f95777800p-2, -0x1.41b6993293ee0p-47,
        0x1.6000000000000p+0, 0x0.0p+0, -0x1.4618bc21c6000p-2, 0x1.3d82f484c84ccp-46,
        0x1.5e00000000000p+0, 0x0.0p+0, -0x1.404308686a800p-2, 0x1.c42f3ed820b3ap-50,
        0x1.5c00000000000p+0, 0x0.0p+0, -0x1.3a64c55694800p-2, 0x1.0b1c686519460p-45,
        0x1.5a00000000000p+0, 0x0.0p+0, -0x1.347dd9a988000p-2, 0x1.5594dd4c58092p-45,
        0x1.5800000000000p+0, 0x0.0p+0, -0x1.2e8e2bae12000p-2, 0x1.67b1e99b72bd8p-45,
        0x1.5600000000000p+0, 0x0.0p+0, -0x1.2895a13de8800p-2, 0x1.5ca14b6cfb03fp-46,
        0x1.5600000000000p+0, 0x0.0p+0, -0x1.2895a13de8800p-2, 0x1.5ca14b6cfb03fp-46,
        0x1.5400000000000p+0, 0x0.0p+0, -0x1.22941fbcf7800p-2, -0x1.65a242853da76p-46,
        0x1.5200000000000p+0, 0x0.0p+0, -0x1.1c898c1699800p-2, -0x1.fafbc68e75404p-46,
        0x1.5000000000000p+0, 0x0.0p+0, -0x1.1675cababa800p-2, 0x1.f1fc63382a8f0p-46,
        0x1.4e00000000000p+0, 0x0.0p+0, -0x1.1058bf9ae4800p-2, -0x1.6a8c4fd055a66p-45,
        0x1.4c00000000000p+0, 0x0.0p+0, -0x1.0a324e2739000p-2, -0x1.c6bee7ef4030ep-47,
        0x1.4a00000000000p+0, 0x0.0p+0, -0x1.0402594b4d000p-2, -0x1.036b89ef42d7fp-48,
        0x1.4a00000000000p+0, 0x0.0p+0, -0x1.0402594b4d000p-2, -0x1.036b89ef42d7fp-48,
        0x1.4800000000000p+0, 0x0.0p+0, -0x1.fb9186d5e4000p-3, 0x1.d572aab993c87p-47,
        0x1.4600000000000p+0, 0x0.0p+0, -0x1.ef0adcbdc6000p-3, 0x1.b26b79c86af24p-45,
        0x1.4400000000000p+0, 0x0.0p+0, -0x1.e27076e2af000p-3, -0x1.72f4f543fff10p-46,
        0x1.4200000000000p+0, 0x0.0p+0, -0x1.d5c216b4fc000p-3, 0x1.1ba91bbca681bp-45,
        0x1.4000000000000p+0, 0x0.0p+0, -0x1.c8ff7c79aa000p-3, 0x1.7794f689f8434p-45,
        0x1.4000000000000p+0, 0x0.0p+0, -0x1.c8ff7c79aa000p-3, 0x1.7794f689f8434p-45,
        0x1.3e00000000000p+0, 0x0.0p+0, -0x1.bc286742d9000p-3, 0x1.94eb0318bb78fp-46,
        0x1.3c00000000000p+0, 0x0.0p+0, -0x1.af3c94e80c000p-3, 0x1.a4e633fcd9066p-52,
        0x1.3a00000000000p+0, 0x0.0p+0, -0x1.a23bc1fe2b000p-3, -0x1.58c64dc46c1eap-45,
        0x1.3a00000000000p+0, 0x0.0p+0, -0x1.a23bc1fe2b000p-3, -0x1.58c64dc46c1eap-45,
        0x1.3800000000000p+0, 0x0.0p+0, -0x1.9525a9cf45000p-3, -0x1.ad1d904c1d4e3p-45,
        0x1.3600000000000p+0, 0x0.0p+0, -0x1.87fa06520d000p-3, 0x1.bbdbf7fdbfa09p-45,
        0x1.3400000000000p+0, 0x0.0p+0, -0x1.7ab890210e000p-3, 0x1.bdb9072534a58p-45,
        0x1.3400000000000p+0, 0x0.0p+0, -0x1.7ab890210e000p-3, 0x1.bdb9072534a58p-45,
        0x1.3200000000000p+0, 0x0.0p+0, -0x1.6d60fe719d000p-3, -0x1.0e46aa3b2e266p-46,
        0x1.3000000000000p+0, 0x0.0p+0, -0x1.5ff3070a79000p-3, -0x1.e9e439f105039p-46,
        0x1.3000000000000p+0, 0x0.0p+0, -0x1.5ff3070a79000p-3, -0x1.e9e439f105039p-46,
        0x1.2e00000000000p+0, 0x0.0p+0, -0x1.526e5e3a1b000p-3, -0x1.0de8b90075b8fp-45,
        0x1.2c00000000000p+0, 0x0.0p+0, -0x1.44d2b6ccb8000p-3, 0x1.70cc16135783cp-46,
        0x1.2c00000000000p+0, 0x0.0p+0, -0x1.44d2b6ccb8000p-3, 0x1.70cc16135783cp-46,
        0x1.2a00000000000p+0, 0x0.0p+0, -0x1.371fc201e9000p-3, 0x1.178864d27543ap-48,
        0x1.2800000000000p+0, 0x0.0p+0, -0x1.29552f81ff000p-3, -0x1.48d301771c408p-45,
        0x1.2600000000000p+0, 0x0.0p+0, -0x1.1b72ad52f6000p-3, -0x1.e80a41811a396p-45,
        0x1.2600000000000p+0, 0x0.0p+0, -0x1.1b72ad52f6000p-3, -0x1.e80a41811a396p-45,
        0x1.2400000000000p+0, 0x0.0p+0, -0x1.0d77e7cd09000p-3, 0x1.a699688e85bf4p-47,
        0x1.2400000000000p+0, 0x0.0p+0, -0x1.0d77e7cd09000p-3, 0x1.a699688e85bf4p-47,
        0x1.2200000000000p+0, 0x0.0p+0, -0x1.fec9131dbe000p-4, -0x1.575545ca333f2p-45,
        0x1.2000000000000p+0, 0x0.0p+0, -0x1.e27076e2b0000p-4, 0x1.a342c2af0003cp-45,
        0x1.2000000000000p+0, 0x0.0p+0, -0x1.e27076e2b0000p-4, 0x1.a342c2af0003cp-45,
        0x1.1e00000000000p+0, 0x0.0p+0, -0x1.c5e548f5bc000p-4, -0x1.d0c57585fbe06p-46,
        0x1.1c00000000000p+0, 0x0.0p+0, -0x1.a926d3a4ae000p-4, 0x1.53935e85baac8p-45,
        0x1.1c00000000000p+0, 0x0.0p+0, -0x1.a926d3a4ae000p-4, 0x1.53935e85baac8p-45,
        0x1.1a00000000000p+0, 0x0.0p+0, -0x1.8c345d631a000p-4, 0x1.37c294d2f5668p-46,
        0x1.1a00000000000p+0, 0x0.0p+0, -0x1.8c345d631a000p-4, 0x1.37c294d2f5668p-46,
        0x1.1800000000000p+0, 0x0.0p+0, -0x1.6f0d28ae56000p-4, -0x1.69737c93373dap-45,
        0x1.1600000000000p+0, 0x0.0p+0, -0x1.51b073f062000p-4, 0x1.f025b61c65e57p-46,
        0x1.1600000000000p+0, 0x0.0p+0, -0x1.51b073f062000p-4, 0x1.f025b61c65e57p-46,
        0x1.1400000000000p+0, 0x0.0p+0, -0x1.341d7961be000p-4, 0x1.c5edaccf913dfp-45,
        0x1.1400000000000p+0, 0x0.0p+0, -0x1.341d7961be000p-4, 0x1.c5edaccf913dfp-45,
        0x1.1200000000000p+0, 0x0.0p+0, -0x1.16536eea38000p-4, 0x1.47c5e768fa309p-46,
        0x1.1000000000000p+0, 0x0.0p+0, -0x1.f0a30c0118000p-5, 0x1.d599e83368e91p-45,
        0x1.1000000000000p+0, 0x0.0p+0, -0x1.f0a30c0118000p-5, 0x1.d599e83368e91p-45,
        0x1.0e00000000000p+0, 0x0.0p+0, -0x1.b42dd71198000p-5, 0x1.c827ae5d6704cp-46,
        0x1.0e00000000000p+0, 0x0.0p+0, -0x1.b42dd71198000p-5, 0x1.c827ae5d6704cp-46,
        0x1.0c00000000000p+0, 0x0.0p+0, -0x1.77458f632c000p-5, -0x1.cfc4634f2a1eep-45,
        0x1.0c00000000000p+0, 0x0.0p+0, -0x1.77458f632c000p-5, -0x1.cfc4634f2a1eep-45,
        0x1.0a00000000000p+0, 0x0.0p+0, -0x1.39e87b9fec000p-5, 0x1.502b7f526feaap-48,
        0x1.0a00000000000p+0, 0x0.0p+0, -0x1.39e87b9fec000p-5, 0x1.502b7f526feaap-48,
        0x1.0800000000000p+0, 0x0.0p+0, -0x1.f829b0e780000p-6, -0x1.980267c7e09e4p-45,
        0x1.0800000000000p+0, 0x0.0p+0, -0x1.f829b0e780000p-6, -0x1.980267c7e09e4p-45,
        0x1.0600000000000p+0, 0x0.0p+0, -0x1.7b91b07d58000p-6, -0x1.88d5493faa639p-45,
        0x1.0400000000000p+0, 0x0.0p+0, -0x1.fc0a8b0fc0000p-7, -0x1.f1e7cf6d3a69cp-50,
        0x1.0400000000000p+0, 0x0.0p+0, -0x1.fc0a8b0fc0000p-7, -0x1.f1e7cf6d3a69cp-50,
        0x1.0200000000000p+0, 0x0.0p+0, -0x1.fe02a6b100000p-8, -0x1.9e23f0dda40e4p-46,
        0x1.0200000000000p+0, 0x0.0p+0, -0x1.fe02a6b100000p-8, -0x1.9e23f0dda40e4p-46,
        0x1.0000000000000p+0, 0x0.0p+0, 0x0.0p+0, 0x0.0p+0,
        0x1.0000000000000p+0, 0x0.0p+0, 0x0.0p+0, 0x0.0p+0,
        0x1.fc00000000000p-1, 0x0.0p+0, 0x1.0101575890000p-7, -0x1.0c76b999d2be8p-46,
        0x1.f800000000000p-1, 0x0.0p+0, 0x1.0205658938000p-6, -0x1.3dc5b06e2f7d2p-45,
        0x1.f400000000000p-1, 0x0.0p+0, 0x1.8492528c90000p-6, -0x1.aa0ba325a0c34p-45,
        0x1.f000000000000p-1, 0x0.0p+0, 0x1.0415d89e74000p-5, 0x1.111c05cf1d753p-47,
        0x1.ec00000000000p-1, 0x0.0p+0, 0x1.466aed42e0000p-5, -0x1.c167375bdfd28p-45,
        0x1.e800000000000p-1, 0x0.0p+0, 0x1.894aa149fc000p-5, -0x1.97995d05a267dp-46,
        0x1.e400000000000p-1, 0x0.0p+0, 0x1.ccb73cdddc000p-5, -0x1.a68f247d82807p-46,
        0x1.e200000000000p-1, 0x0.0p+0, 0x1.eea31c006c000p-5, -0x1.e113e4fc93b7bp-47,
        0x1.de00000000000p-1, 0x0.0p+0, 0x1.1973bd1466000p-4, -0x1.5325d560d9e9bp-45,
        0x1.da00000000000p-1, 0x0.0p+0, 0x1.3bdf5a7d1e000p-4, 0x1.cc85ea5db4ed7p-45,
        0x1.d600000000000p-1, 0x0.0p+0, 0x1.5e95a4d97a000p-4, -0x1.c69063c5d1d1ep-45,
        0x1.d400000000000p-1, 0x0.0p+0, 0x1.700d30aeac000p-4, 0x1.c1e8da99ded32p-49,
        0x1.d000000000000p-1, 0x0.0p+0, 0x1.9335e5d594000p-4, 0x1.3115c3abd47dap-45,
        0x1.cc00000000000p-1, 0x0.0p+0, 0x1.b6ac88dad6000p-4, -0x1.390802bf768e5p-46,
        0x1.ca00000000000p-1, 0x0.0p+0, 0x1.c885801bc4000p-4, 0x1.646d1c65aacd3p-45,
        0x1.c600000000000p-1, 0x0.0p+0, 0x1.ec739830a2000p-4, -0x1.dc068afe645e0p-45,
        0x1.c400000000000p-1, 0x0.0p+0, 0x1.fe89139dbe000p-4, -0x1.534d64fa10afdp-45,
        0x1.c000000000000p-1, 0x0.0p+0, 0x1.1178e8227e000p-3, 0x1.1ef78ce2d07f2p-45,
        0x1.be00000000000p-1, 0x0.0p+0, 0x1.1aa2b7e23f000p-3, 0x1.ca78e44389934p-45,
        0x1.ba00000000000p-1, 0x0.0p+0, 0x1.2d1610c868000p-3, 0x1.39d6ccb81b4a1p-47,
        0x1.b800000000000p-1, 0x0.0p+0, 0x1.365fcb0159000p-3, 0x1.62fa8234b7289p-51,
        0x1.b400000000000p-1, 0x0.0p+0, 0x1.4913d8333b000p-3, 0x1.5837954fdb678p-45,
        0x1.b200000000000p-1, 0x0.0p+0, 0x1.527e5e4a1b000p-3, 0x1.633e8e5697dc7p-45,
        0x1.ae00000000000p-1, 0x0.0p+0, 0x1.6574ebe8c1000p-3, 0x1.9cf8b2c3c2e78p-46,
        0x1.ac00000000000p-1, 0x0.0p+0, 0x1.6f0128b757000p-3, -0x1.5118de59c21e1p-45,
        0x1.aa00000000000p-1, 0x0.0p+0, 0x1.7898d85445000p-3, -0x1.c661070914305p-46,
        0x1.a600000000000p-1, 0x0.0p+0, 0x1.8beafeb390000p-3, -0x1.73d54aae92cd1p-47,
        0x1.a400000000000p-1, 0x0.0p+0, 0x1.95a5adcf70000p-3, 0x1.7f22858a0ff6fp-47,
        0x1.a000000000000p-1, 0x0.0p+0, 0x1.a93ed3c8ae000p-3, -0x1.8724350562169p-45,
        0x1.9e00000000000p-1, 0x0.0p+0, 0x1.b31d8575bd000p-3, -0x1.c358d4eace1aap-47,
        0x1.9c00000000000p-1, 0x0.0p+0, 0x1.bd087383be000p-3, -0x1.d4bc4595412b6p-45,
        0x1.9a00000000000p-1, 0x0.0p+0, 0x1.c6ffbc6f01000p-3, -0x1.1ec72c5962bd2p-48,
        0x1.9600000000000p-1, 0x0.0p+0, 0x1.db13db0d49000p-3, -0x1.aff2af715b035p-45,
        0x1.9400000000000p-1, 0x0.0p+0, 0x1.e530effe71000p-3, 0x1.212276041f430p-51,
        0x1.9200000000000p-1, 0x0.0p+0, 0x1.ef5ade4dd0000p-3, -0x1.a211565bb8e11p-51,
        0x1.9000000000000p-1, 0x0.0p+0, 0x1.f991c6cb3b000p-3, 0x1.bcbecca0cdf30p-46,
        0x1.8c00000000000p-1, 0x0.0p+0, 0x1.07138604d5800p-2, 0x1.89cdb16ed4e91p-48,
        0x1.8a00000000000p-1, 0x0.0p+0, 0x1.0c42d67616000p-2, 0x1.7188b163ceae9p-45,
        0x1.8800000000000p-1, 0x0.0p+0, 0x1.1178e8227e800p-2, -0x1.c210e63a5f01cp-45,
        0x1.8600000000000p-1, 0x0.0p+0, 0x1.16b5ccbacf800p-2, 0x1.b9acdf7a51681p-45,
        0x1.8400000000000p-1, 0x0.0p+0, 0x1.1bf99635a6800p-2, 0x1.ca6ed5147bdb7p-45,
        0x1.8200000000000p-1, 0x0.0p+0, 0x1.214456d0eb800p-2, 0x1.a87deba46baeap-47,
        0x1.7e00000000000p-1, 0x0.0p+0, 0x1.2bef07cdc9000p-2, 0x1.a9cfa4a5004f4p-45,
        0x1.7c00000000000p-1, 0x0.0p+0, 0x1.314f1e1d36000p-2, -0x1.8e27ad3213cb8p-45,
        0x1.7a00000000000p-1, 0x0.0p+0, 0x1.36b6776be1000p-2, 0x1.16ecdb0f177c8p-46,
        0x1.7800000000000p-1, 0x0.0p+0, 0x1.3c25277333000p-2, 0x1.83b54b606bd5cp-46,
        0x1.7600000000000p-1, 0x0.0p+0, 0x1.419b423d5e800p-2, 0x1.8e436ec90e09dp-47,
        0x1.7400000000000p-1, 0x0.0p+0, 0x1.4718dc271c800p-2, -0x1.f27ce0967d675p-45,
        0x1.7200000000000p-1, 0x0.0p+0, 0x1.4c9e09e173000p-2, -0x1.e20891b0ad8a4p-45,
        0x1.7000000000000p-1, 0x0.0p+0, 0x1.522ae0738a000p-2, 0x1.ebe708164c759p-45,
        0x1.6e00000000000p-1, 0x0.0p+0, 0x1.57bf753c8d000p-2, 0x1.fadedee5d40efp-46,
        0x1.6c00000000000p-1, 0x0.0p+0, 0x1.5d5bddf596000p-2, -0x1.a0b2a08a465dcp-47,
    };
    return tab;
}
// 2^(i / 128), i = 0 ... 127, as {tail, bits of the value minus i << 45} -- __exp_data.tab (e_exp_data.c)
RDR_FN const uint64_t *exp_table() {
    static const uint64_t tab[256] = {
        0x0000000000000000ull, 0x3ff0000000000000ull, 0x3c9b3b4f1a88bf6eull, 0x3feff63da9fb3335ull,
        0xbc7160139cd8dc5dull, 0x3fefec9a3e778061ull, 0xbc905e7a108766d1ull, 0x3fefe315e86e7f85ull,
        0x3c8cd2523567f613ull, 0x3fefd9b0d3158574ull, 0xbc8bce8023f98efaull, 0x3fefd06b29ddf6deull,
        0x3c60f74e61e6c861ull, 0x3fefc74518759bc8ull, 0x3c90a3e45b33d399ull, 0x3fefbe3ecac6f383ull,
        0x3c979aa65d837b6dull, 0x3fefb5586cf9890full, 0x3c8eb51a92fdeffcull, 0x3fefac922b7247f7ull,
        0x3c3ebe3d702f9cd1ull, 0x3fefa3ec32d3d1a2ull, 0xbc6a033489906e0bull, 0x3fef9b66affed31bull,
        0xbc9556522a2fbd0eull, 0x3fef9301d0125b51ull, 0xbc5080ef8c4eea55ull, 0x3fef8abdc06c31ccull,
        0xbc91c923b9d5f416ull, 0x3fef829aaea92de0ull, 0x3c80d3e3e95c55afull, 0x3fef7a98c8a58e51ull,
        0xbc801b15eaa59348ull, 0x3fef72b83c7d517bull, 0xbc8f1ff055de323dull, 0x3fef6af9388c8deaull,
        0x3c8b898c3f1353bfull, 0x3fef635beb6fcb75ull, 0xbc96d99c7611eb26ull, 0x3fef5be084045cd4ull,
        0x3c9aecf73e3a2f60ull, 0x3fef54873168b9aaull, 0xbc8fe782cb86389dull, 0x3fef4d5022fcd91dull,
        0x3c8a6f4144a6c38dull, 0x3fef463b88628cd6ull, 0x3c807a05b0e4047dull, 0x3fef3f49917ddc96ull,
        0x3c968efde3a8a894ull, 0x3fef387a6e756238ull, 0x3c875e18f274487dull, 0x3fef31ce4fb2a63full,
        0x3c80472b981fe7f2ull, 0x3fef2b4565e27cddull, 0xbc96b87b3f71085eull, 0x3fef24dfe1f56381ull,
        0x3c82f7e16d09ab31ull, 0x3fef1e9df51fdee1ull, 0xbc3d219b1a6fbffaull, 0x3fef187fd0dad990ull,
        0x3c8b3782720c0ab4ull, 0x3fef1285a6e4030bull, 0x3c6e149289cecb8full, 0x3fef0cafa93e2f56ull,
        0x3c834d754db0abb6ull, 0x3fef06fe0a31b715ull, 0x3c864201e2ac744cull, 0x3fef0170fc4cd831ull,
        0x3c8fdd395dd3f84aull, 0x3feefc08b26416ffull, 0xbc86a3803b8e5b04ull, 0x3feef6c55f929ff1ull,
        0xbc924aedcc4b5068ull, 0x3feef1a7373aa9cbull, 0xbc9907f81b512d8eull, 0x3feeecae6d05d866ull,
        0xbc71d1e83e9436d2ull, 0x3feee7db34e59ff7ull, 0xbc991919b3ce1b15ull, 0x3feee32dc313a8e5ull,
        0x3c859f48a72a4c6dull, 0x3feedea64c123422ull, 0xbc9312607a28698aull, 0x3feeda4504ac801cull,
        0xbc58a78f4817895bull, 0x3feed60a21f72e2aull, 0xbc7c2c9b67499a1bull, 0x3feed1f5d950a897ull,
        0x3c4363ed60c2ac11ull, 0x3feece086061892dull, 0x3c9666093b0664efull, 0x3feeca41ed1d0057ull,
        0x3c6ecce1daa10379ull, 0x3feec6a2b5c13cd0ull, 0x3c93ff8e3f0f1230ull, 0x3feec32af0d7d3deull,
        0x3c7690cebb7aafb0ull, 0x3feebfdad5362a27ull, 0x3c931dbdeb54e077ull, 0x3feebcb299fddd0dull,
        0xbc8f94340071a38eull, 0x3feeb9b2769d2ca7ull, 0xbc87deccdc93a349ull, 0x3feeb6daa2cf6642ull,
        0xbc78dec6bd0f385full, 0x3feeb42b569d4f82ull, 0xbc861246ec7b5cf6ull, 0x3feeb1a4ca5d920full,
        0x3c93350518fdd78eull, 0x3feeaf4736b527daull, 0x3c7b98b72f8a9b05ull, 0x3feead12d497c7fdull,
        0x3c9063e1e21c5409ull, 0x3feeab07dd485429ull, 0x3c34c7855019c6eaull, 0x3feea9268a5946b7ull,
        0x3c9432e62b64c035ull, 0x3feea76f15ad2148ull, 0xbc8ce44a6199769full, 0x3feea5e1b976dc09ull,
        0xbc8c33c53bef4da8ull, 0x3feea47eb03a5585ull, 0xbc845378892be9aeull, 0x3feea34634ccc320ull,
        0xbc93cedd78565858ull, 0x3feea23882552225ull, 0x3c5710aa807e1964ull, 0x3feea155d44ca973ull,
        0xbc93b3efbf5e2228ull, 0x3feea09e667f3bcdull, 0xbc6a12ad8734b982ull, 0x3feea012750bdabfull,
        0xbc6367efb86da9eeull, 0x3fee9fb23c651a2full, 0xbc80dc3d54e08851ull, 0x3fee9f7df9519484ull,
        0xbc781f647e5a3ecfull, 0x3fee9f75e8ec5f74ull, 0xbc86ee4ac08b7db0ull, 0x3fee9f9a48a58174ull,
        0xbc8619321e55e68aull, 0x3fee9feb564267c9ull, 0x3c909ccb5e09d4d3ull, 0x3feea0694fde5d3full,
        0xbc7b32dcb94da51dull, 0x3feea11473eb0187ull, 0x3c94ecfd5467c06bull, 0x3feea1ed0130c132ull,
        0x3c65ebe1abd66c55ull, 0x3feea2f336cf4e62ull, 0xbc88a1c52fb3cf42ull, 0x3feea427543e1a12ull,
        0xbc9369b6f13b3734ull, 0x3feea589994cce13ull, 0xbc805e843a19ff1eull, 0x3feea71a4623c7adull,
        0xbc94d450d872576eull, 0x3feea8d99b4492edull, 0x3c90ad675b0e8a00ull, 0x3feeaac7d98a6699ull,
        0x3c8db72fc1f0eab4ull, 0x3feeace5422aa0dbull, 0xbc65b6609cc5e7ffull, 0x3feeaf3216b5448cull,
        0x3c7bf68359f35f44ull, 0x3feeb1ae99157736ull, 0xbc93091fa71e3d83ull, 0x3feeb45b0b91ffc6ull,
        0xbc5da9b88b6c1e29ull, 0x3feeb737b0cdc5e5ull, 0xbc6c23f97c90b959ull, 0x3feeba44cbc8520full,
        0xbc92434322f4f9aaull, 0x3feebd829fde4e50ull, 0xbc85ca6cd7668e4bull, 0x3feec0f170ca07baull,
        0x3c71affc2b91ce27ull, 0x3feec49182a3f090ull, 0x3c6dd235e10a73bbull, 0x3feec86319e32323ull,
        0xbc87c50422622263ull, 0x3feecc667b5de565ull, 0x3c8b1c86e3e231d5ull, 0x3feed09bec4a2d33ull,
        0xbc91bbd1d3bcbb15ull, 0x3feed503b23e255dull, 0x3c90cc319cee31d2ull, 0x3feed99e1330b358ull,
        0x3c8469846e735ab3ull, 0x3feede6b5579fdbfull, 0xbc82dfcd978e9db4ull, 0x3feee36bbfd3f37aull,
        0x3c8c1a7792cb3387ull, 0x3feee89f995ad3adull, 0xbc907b8f4ad1d9faull, 0x3feeee07298db666ull,
        0xbc55c3d956dcaebaull, 0x3feef3a2b84f15fbull, 0xbc90a40e3da6f640ull, 0x3feef9728de5593aull,
        0xbc68d6f438ad9334ull, 0x3feeff76f2fb5e47ull, 0xbc91eee26b588a35ull, 0x3fef05b030a1064aull,
        0x3c74ffd70a5fddcdull, 0x3fef0c1e904bc1d2ull, 0xbc91bdfbfa9298acull, 0x3fef12c25bd71e09ull,
        0x3c736eae30af0cb3ull, 0x3fef199bdd85529cull, 0x3c8ee3325c9ffd94ull, 0x3fef20ab5fffd07aull,
        0x3c84e08fd10959acull, 0x3fef27f12e57d14bull, 0x3c63cdaf384e1a67ull, 0x3fef2f6d9406e7b5ull,
        0x3c676b2c6c921968ull, 0x3fef3720dcef9069ull, 0xbc808a1883ccb5d2ull, 0x3fef3f0b555dc3faull,
        0xbc8fad5d3ffffa6full, 0x3fef472d4a07897cull, 0xbc900dae3875a949ull, 0x3fef4f87080d89f2ull,
        0x3c74a385a63d07a7ull, 0x3fef5818dcfba487ull, 0xbc82919e2040220full, 0x3fef60e316c98398ull,
        0x3c8e5a50d5c192acull, 0x3fef69e603db3285ull, 0x3c843a59ac016b4bull, 0x3fef7321f301b460ull,
        0xbc82d52107b43e1full, 0x3fef7c97337b9b5full, 0xbc892ab93b470dc9ull, 0x3fef864614f5a129ull,
        0x3c74b604603a88d3ull, 0x3fef902ee78b3ff6ull, 0x3c83c5ec519d7271ull, 0x3fef9a51fbc74c83ull,
        0xbc8ff7128fd391f0ull, 0x3fefa4afa2a490daull, 0xbc8dae98e223747dull, 0x3fefaf482d8e67f1ull,
        0x3c8ec3bc41aa2008ull, 0x3fefba1bee615a27ull, 0x3c842b94c3a9eb32ull, 0x3fefc52b376bba97ull,
        0x3c8a64a931d185eeull, 0x3fefd0765b6e4540ull, 0xbc8e37bae43be3edull, 0x3fefdbfdad9cbe14ull,
        0x3c77893b4d91cd9dull, 0x3fefe7c1819e90d8ull, 0x3c5305c14160cc89ull, 0x3feff3c22b8f71f1ull,
    };
    return tab;
}

// e_log.c: __log
RDR_FN double log(double x) {
    uint64_t ix = d2bits(x);
    const uint32_t top = (uint32_t)(ix >> 48);
    if (ix - 0x3fee000000000000ull < 0x3090000000000ull) {             // 1 - 2^-4 <= x < 1 + 0x1.09p-4
        if (ix == 0x3ff0000000000000ull) return 0.0;
        const double B0 = -0x1p-1, B1 = 0x1.5555555555577p-2, B2 = -0x1.ffffffffffdcbp-3, B3 = 0x1.999999995dd0cp-3,
                     B4 = -0x1.55555556745a7p-3, B5 = 0x1.24924a344de30p-3, B6 = -0x1.fffffa4423d65p-4, B7 = 0x1.c7184282ad6cap-4,
                     B8 = -0x1.999eb43b068ffp-4, B9 = 0x1.78182f7afd085p-4, B10 = -0x1.5521375d145cdp-4;
        const double r = x - 1.0;
        const double r2 = r * r, r3 = r * r2;
        double p = fma(r3, B10, fma(r2, B9, fma(r, B8, B7)));
        p = fma(p, r3, fma(r2, B6, fma(r, B5, B4)));
        p = fma(p, r3, fma(r2, B3, fma(r, B2, B1)));
        const double rhi = fma(-0x1p27, r, fma(r, 0x1p27, r));
        const double rlo = r - rhi;
        const double rr = rhi * rhi;
        const double hi = fma(rr, B0, r);
        double lo = fma(rr, B0, r - hi);
        lo = fma(B0 * rlo, rhi + r, lo);
        return hi + fma(p, r3, lo);
    }
    if (top - 0x0010u >= 0x7ff0u - 0x0010u) {
        if (ix * 2 == 0) return -INFINITY;
        if (ix == 0x7ff0000000000000ull) return x;
        if ((top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u) return (x - x) / (x - x);
        ix = d2bits(x * 0x1p52) - (52ull << 52);                       // subnormal
    }
    const double Ln2hi = 0x1.62e42fefa3800p-1, Ln2lo = 0x1.ef35793c76730p-45, A0 = -0x1.0000000000001p-1, A1 = 0x1.555555551305bp-2,
                 A2 = -0x1.fffffffeb4590p-3, A3 = 0x1.999b324f10111p-3, A4 = -0x1.55575e506c89fp-3;
    const uint64_t tmp = ix - 0x3fe6000000000000ull;
    const int i = (int)((tmp >> 45) & 127);
    const int k = (int)((int64_t)tmp >> 52);
    const double z = bits2d(ix - (tmp & (0xfffull << 52)));
    const double invc = log_table()[2 * i], logc = log_table()[2 * i + 1];
    const double r = fma(z, invc, -1.0);
    const double kd = (double)k;
    const double w = fma(kd, Ln2hi, logc);
    const double hi = w + r;
    const double lo = fma(kd, Ln2lo, (w - hi) + r);
    const double r2 = r * r;
    const double q = fma(fma(r, A4, A3), r2, fma(r, A2, A1));
    return fma(r * r2, q, fma(r2, A0, lo)) + hi;
}

namespace detail {
// e_pow.c: checkint -- 0: not an integer, 1: odd, 2: even
RDR_FN int pow_checkint(uint64_t iy) {
    const int e = (int)(iy >> 52 & 0x7ff);
    if (e < 0x3ff) return 0;
    if (e > 0x3ff + 52) return 2;
    if (iy & ((1ull << (0x3ff + 52 - e)) - 1)) return 0;
    if (iy & (1ull << (0x3ff + 52 - e))) return 1;
    return 2;
}
RDR_FN bool pow_zeroinfnan(uint64_t i) { return 2 * i - 1 >= 2 * 0x7ff0000000000000ull - 1; }
RDR_FN bool pow_issignaling(uint64_t ix) { return 2 * (ix ^ 0x0008000000000000ull) > 2 * 0x7ff8000000000000ull; }
// e_pow.c: exp_inline -- sign * exp(x + xtail)
RDR_FN double pow_exp(double x, double xtail, uint32_t sign_bias) {
    const double InvLn2N = 0x1.71547652b82fep+7, Shift = 0x1.8p+52, NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47,
                 C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
    uint32_t abstop = (uint32_t)(d2bits(x) >> 52) & 0x7ff;
    if (abstop - 0x3c9u >= 0x3fu) {
        if (abstop - 0x3c9u >= 0x80000000u) { const double one = 1.0 + x; return sign_bias ? -one : one; }
        if (abstop >= 0x409u) {
            if (d2bits(x) >> 63) return sign_bias ? -0.0 : 0.0;
            return sign_bias ? -INFINITY : INFINITY;
        }
        abstop = 0;
    }
    double kd = fma(x, InvLn2N, Shift);
    const uint64_t ki = d2bits(kd);
    kd -= Shift;
    double r = fma(kd, NegLn2loN, fma(kd, NegLn2hiN, x));
    r = xtail + r;
    const uint64_t idx = 2 * (ki % 128);
    const uint64_t top = (ki + sign_bias) << 45;
    const double tail = bits2d(exp_table()[idx]);
    uint64_t sbits = exp_table()[idx + 1] + top;
    const double r2 = r * r;
    const double tmp = fma(fma(r, C5, C4), r2 * r2, fma(fma(r, C3, C2), r2, r + tail));
    if (abstop == 0) {                                                   // e_pow.c: specialcase
        if ((ki & 0x80000000u) == 0) {
            sbits -= 1009ull << 52;
            const double scale = bits2d(sbits);
            return 0x1p1009 * fma(scale, tmp, scale);
        }
        sbits += 1022ull << 52;
        const double scale = bits2d(sbits);
        const double st = tmp * scale;
        double y = scale + st;
        if (fabs(y) < 1.0) {
            const double one = (y < 0.0) ? -1.0 : 1.0;
            double lo = (scale - y) + st;
            const double hi = y + one;
            lo = ((one - hi) + y) + lo;
            y = (lo + hi) - one;
            if (y == 0.0) y = bits2d(sbits & 0x8000000000000000ull);
        }
        return 0x1p-1022 * y;
    }
    const double scale = bits2d(sbits);
    return fma(tmp, scale, scale);
}
} // namespace detail

// e_pow.c: __pow
RDR_FN double pow(double x, double y) {
    using namespace detail;
    uint32_t sign_bias = 0;
    uint64_t ix = d2bits(x);
    const uint64_t iy = d2bits(y);
    uint32_t topx = (uint32_t)(ix >> 52);
    const uint32_t topy = (uint32_t)(iy >> 52);
    const uint64_t one_bits = 0x3ff0000000000000ull, inf_bits = 0x7ff0000000000000ull;
    if (topx - 0x001u >= 0x7ffu - 0x001u || (topy & 0x7ff) - 0x3beu >= 0x43eu - 0x3beu) {
        if (pow_zeroinfnan(iy)) {
            if (2 * iy == 0) return pow_issignaling(ix) ? x + y : 1.0;
            if (ix == one_bits) return pow_issignaling(iy) ? x + y : 1.0;
            if (2 * ix > 2 * inf_bits || 2 * iy > 2 * inf_bits) return x + y;
            if (2 * ix == 2 * one_bits) return 1.0;
            if ((2 * ix < 2 * one_bits) == !(iy >> 63)) return 0.0;
            return y * y;
        }
        if (pow_zeroinfnan(ix)) {
            double x2 = x * x;
            if ((ix >> 63) && pow_checkint(iy) == 1) { x2 = -x2; sign_bias = 1; }
            if (2 * ix == 0 && (iy >> 63)) return sign_bias ? -INFINITY : INFINITY;
            return (iy >> 63) ? 1 / x2 : x2;
        }
        if (ix >> 63) {
            const int yint = pow_checkint(iy);
            if (yint == 0) return (x - x) / (x - x);
            if (yint == 1) sign_bias = 0x800u << 7;
            ix &= 0x7fffffffffffffffull;
            topx &= 0x7ff;
        }
        if ((topy & 0x7ff) - 0x3beu >= 0x43eu - 0x3beu) {
            if (ix == one_bits) return 1.0;
            if ((topy & 0x7ff) < 0x3be) return ix > one_bits ? 1.0 + y : 1.0 - y;
            return ((ix > one_bits) == (topy < 0x800)) ? INFINITY : 0.0;
        }
        if (topx == 0) {
            ix = d2bits(x * 0x1p52);
            ix &= 0x7fffffffffffffffull;
            ix -= 52ull << 52;
        }
    }
    // e_pow.c: log_inline -- hi + lo = log(x)
    const double Ln2hi = 0x1.62e42fefa3800p-1, Ln2lo = 0x1.ef35793c76730p-45, A0 = -0x1p-1, A1 = -0x1.5555555555560p-1, A2 = 0x1.0000000000006p-1,
                 A3 = 0x1.999999959554ep-1, A4 = -0x1.555555529a47ap-1, A5 = -0x1.2495b9b4845e9p+0, A6 = 0x1.0002b8b263fc3p+0;
    const uint64_t tmp = ix - 0x3fe6955500000000ull;
    const int i = (int)((tmp >> 45) & 127);
    const int k = (int)((int64_t)tmp >> 52);
    const double z = bits2d(ix - (tmp & (0xfffull << 52)));
    const double kd = (double)k;
    const double *e = pow_log_table() + 4 * i;
    const double invc = e[0], logc = e[2], logctail = e[3];
    const double r = fma(z, invc, -1.0);
    const double t1 = fma(kd, Ln2hi, logc);
    const double t2 = r + t1;
    const double lo1 = fma(kd, Ln2lo, logctail);
    const double lo2 = (t1 - t2) + r;
    const double ar = r * A0;
    const double ar2 = r * ar;
    const double ar3 = r * ar2;
    const double hi = t2 + ar2;
    const double lo3 = fma(ar, r, -ar2);
    const double lo4 = (t2 - hi) + ar2;
    const double p = fma(ar2, fma(fma(r, A6, A5), ar2, fma(r, A4, A3)), fma(r, A2, A1));
    const double lo = fma(ar3, p, ((lo1 + lo2) + lo3) + lo4);
    const double lhi = hi + lo;
    const double llo = (hi - lhi) + lo;
    const double ehi = y * lhi;
    const double elo = fma(y, llo, fma(y, lhi, -ehi));
    return pow_exp(ehi, elo, sign_bias);
}

// a float (or integer) argument means another glibc routine in the reference: refuse it
double sin(float) = delete; double cos(float) = delete; double atan(float) = delete; double acos(float) = delete; double log(float) = delete;
double atan2(float, float) = delete; double pow(float, float) = delete; double pow(double, int) = delete; double pow(float, int) = delete;

} // namespace gm
