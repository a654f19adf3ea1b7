// gpt_scene.hip.h -- the scene handle behind include/gdpt_tracer.h, shared by the translation units of the library (gpt_capi.hip builds and owns
// it; gbdpt_capi.hip renders from it).
#pragma once
#include "gpt_kernels.hip.h"
#include <vector>

struct gdpt_scene {
    gdpt_tr::SceneD d;
    std::vector<void *> allocs;
    int device = 0;
    int bvhDepth = 0;
    int numCUs = 256;
    bool specialEmitters = false;   // an environment or point emitter: the ENV builds of the render kernel
    bool hittableEmitters = true;   // some emitter is not a point (G-BDPT: the sensor subpath's extra step, gbdpt_proc.cpp:120-122)
    bool perVertex = false;         // vertex normals or bitmap textures: the builds that keep a hit's barycentrics
    size_t ldsSceneBytes = 0;
    double bsphereRadius = 0.0;     // Scene::getBSphere().radius after Scene::initializeBidirectional (kd-tree bounds + sensor + emitters, scene.cpp:386-413)
    bool cropped = false;       // the camera's film has a crop window (gdpt_camera::fullWidth > 0): the G-BDPT entry points refuse it
    std::vector<gdpt_tr::MaterialD> hostMats;   // the material table as uploaded (G-BDPT checks its scope against it)
};
