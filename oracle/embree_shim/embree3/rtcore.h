// Stand-in for the 15 Embree-3 entry points the reference calls
// (/root/reference/src/scene.cpp:129-154 build, :556-574 rtcIntersect1, :667-682 rtcOccluded1,
//  :311-312 teardown).  Embree itself (un-vendored submodule redner-dependencies, "embree3",
// no version pin) is not available offline.  This is TEST INFRASTRUCTURE for the parity oracle:
// it is linked only into oracle/_ref/ and never into the product.
//
// Semantics: exact closest hit under the fp32 predicate of redner_amd/csrc/raytri.h
// (smallest t in (tnear, tfar), ties -> smaller (geomID, primID)); any-hit for rtcOccluded1.
// geomID = attach order, primID = triangle index, as in Embree.
#pragma once
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct RTCDeviceTy   *RTCDevice;
typedef struct RTCSceneTy    *RTCScene;
typedef struct RTCGeometryTy *RTCGeometry;

#define RTC_INVALID_GEOMETRY_ID ((unsigned int)-1)

enum RTCBuildQuality { RTC_BUILD_QUALITY_LOW = 0, RTC_BUILD_QUALITY_MEDIUM = 1, RTC_BUILD_QUALITY_HIGH = 2 };
enum RTCSceneFlags { RTC_SCENE_FLAG_NONE = 0, RTC_SCENE_FLAG_DYNAMIC = 1, RTC_SCENE_FLAG_COMPACT = 2,
                     RTC_SCENE_FLAG_ROBUST = 4 };
enum RTCGeometryType { RTC_GEOMETRY_TYPE_TRIANGLE = 0 };
enum RTCBufferType { RTC_BUFFER_TYPE_INDEX = 0, RTC_BUFFER_TYPE_VERTEX = 1 };
enum RTCFormat { RTC_FORMAT_UINT3 = 0x5003, RTC_FORMAT_FLOAT3 = 0x9003 };

struct RTCRay {
    float org_x, org_y, org_z, tnear;
    float dir_x, dir_y, dir_z, time;
    float tfar;
    unsigned int mask, id, flags;
};
struct RTCHit {
    float Ng_x, Ng_y, Ng_z, u, v;
    unsigned int primID, geomID, instID[1];
};
struct RTCRayHit {
    struct RTCRay ray;
    struct RTCHit hit;
};
struct RTCIntersectContext {
    unsigned int flags;
    void *filter;
    unsigned int instID[1];
};

RTCDevice rtcNewDevice(const char *config);
void rtcReleaseDevice(RTCDevice);
RTCScene rtcNewScene(RTCDevice);
void rtcReleaseScene(RTCScene);
void rtcSetSceneBuildQuality(RTCScene, enum RTCBuildQuality);
void rtcSetSceneFlags(RTCScene, enum RTCSceneFlags);
RTCGeometry rtcNewGeometry(RTCDevice, enum RTCGeometryType);
void *rtcSetNewGeometryBuffer(RTCGeometry, enum RTCBufferType, unsigned int slot, enum RTCFormat,
                              size_t byte_stride, size_t item_count);
void rtcSetGeometryVertexAttributeCount(RTCGeometry, unsigned int);
void rtcCommitGeometry(RTCGeometry);
unsigned int rtcAttachGeometry(RTCScene, RTCGeometry);
void rtcReleaseGeometry(RTCGeometry);
void rtcCommitScene(RTCScene);
void rtcIntersect1(RTCScene, struct RTCIntersectContext *, struct RTCRayHit *);
void rtcOccluded1(RTCScene, struct RTCIntersectContext *, struct RTCRay *);

static inline void rtcInitIntersectContext(struct RTCIntersectContext *c) {
    c->flags = 0; c->filter = 0; c->instID[0] = RTC_INVALID_GEOMETRY_ID;
}

#ifdef __cplusplus
}
#endif
