// oracle/ref/ref_scene.h — TEST INFRASTRUCTURE ONLY: the state behind the ref_scene_* entry points (the reference's HashParams + HashDataStruct)
#ifndef BF_REF_SCENE_H
#define BF_REF_SCENE_H
struct ref_scene {
    HashParams params;
    HashDataStruct data;
    unsigned int numIntegrated = 0;
};
#endif
