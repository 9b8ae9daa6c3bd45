/*
 * volpath_simple_hip.cpp -- Mitsuba 0.6 integrator plugin `volpath_simple_hip`: the shim of path_hip.cpp (see there) for the sibling integrator
 * `volpath_simple` (src/integrators/path/volpath_simple.cpp) on scenes WITHOUT participating media -- the path tracer without multiple importance sampling,
 * PHIP_INTEGRATOR_VOLPATH_SIMPLE of include/phip.h.  Media are outside the back end's scope: a scene that has any is refused in
 * preprocess() with an error, as the reference refuses what a plugin cannot render (Log(EError)).
 *
 * Build (inside a Mitsuba 0.6 source tree, next to src/integrators/path/):
 *     plugins += env.SharedLibrary('path_hip', ['path_hip/path_hip.cpp'], LIBS = env['LIBS'] + ['phip'])
 * (pattern: src/integrators/SConscript:5).  Select it with <integrator type="path_hip"/>.
 * In this repository it is compiled against the reference's own headers and run inside the reference's libraries
 * (oracle/Makefile.ref, target `shims`; tests/test_gpu_dropin.py); the standalone harness (mitsuba_amd/*.py, tests/)
 * drives the same C ABI through ctypes.  See INTEGRATION.md.
 *
 * What it does, and nothing else:
 *   - derives from MonteCarloIntegrator so that maxDepth / rrDepth / strictNormals / hideEmitters
 *     parse, validate and serialise exactly like `path` (src/librender/integrator.cpp:190-225);
 *   - preprocess(): flattens Scene::getShapes() into a phip_scene_desc (every Shape through
 *     createTriMesh() unless it already is a TriMesh) and calls phip_scene_create (phip_flatten.h, shared with direct_hip.cpp);
 *   - render(): overrides SamplingIntegrator::render (integrator.cpp:95-129): one phip_render call,
 *     then film->put() of one full-frame ImageBlock; returns false when cancelled;
 *   - cancel(): phip_cancel (integrator.cpp:90-93);
 *   - Li(): still required by the interface (integrator.h:321-322, used by `adaptive`/`irrcache`):
 *     delegates to a nested CPU `path` integrator with the same parameters.
 */
#include "phip_flatten.h"

MTS_NAMESPACE_BEGIN

class VolPathSimpleHIP : public MonteCarloIntegrator {
public:
    VolPathSimpleHIP(const Properties &props) : MonteCarloIntegrator(props) {
        m_holder.setDevice(props.getInteger("device", 0));
        m_holder.setDeviceCount(props.getInteger("devices", 1));       /* GPUs of the node to spread the job over (0 = all) */
        Properties p("volpath_simple");
        p.setInteger("maxDepth", m_maxDepth); p.setInteger("rrDepth", m_rrDepth);
        p.setBoolean("strictNormals", m_strictNormals); p.setBoolean("hideEmitters", m_hideEmitters);
        m_cpuPath = static_cast<SamplingIntegrator *>(PluginManager::getInstance()->createObject(MTS_CLASS(Integrator), p));
    }

    VolPathSimpleHIP(Stream *stream, InstanceManager *manager) : MonteCarloIntegrator(stream, manager) {
        m_holder.setDevice(stream->readInt());
        m_holder.setDeviceCount(stream->readInt());
        m_cpuPath = static_cast<SamplingIntegrator *>(manager->getInstance(stream));
    }

    void serialize(Stream *stream, InstanceManager *manager) const {
        MonteCarloIntegrator::serialize(stream, manager);
        stream->writeInt(m_holder.getDevice());
        stream->writeInt(m_holder.getDeviceCount());
        manager->serialize(stream, m_cpuPath.get());
    }

    Spectrum Li(const RayDifferential &ray, RadianceQueryRecord &rRec) const {
        /* reached only through an integrator that wraps this one (`adaptive`, `irrcache`): those call Li() per sample on the host */
        static bool told = false;
        if (!told) { told = true; SLog(EWarn, "volpath_simple_hip: Li() was called by a wrapping integrator -- these samples run on the CPU (nested `volpath_simple`), not on the GPU"); }
        return m_cpuPath->Li(ray, rRec);
    }

    bool preprocess(const Scene *scene, RenderQueue *queue, const RenderJob *job, int sceneResID, int sensorResID, int samplerResID) {
        if (!MonteCarloIntegrator::preprocess(scene, queue, job, sceneResID, sensorResID, samplerResID))
            return false;
        if (!scene->getMedia().empty())
            Log(EError, "volpath_simple_hip: the scene contains participating media -- the GPU back end renders surfaces only (use `volpath_simple`)");
        m_holder.flatten(scene);
        return true;
    }

    bool render(Scene *scene, RenderQueue *queue, const RenderJob *job, int sceneResID, int sensorResID, int samplerResID) {
        phip_render_params rp; memset(&rp, 0, sizeof(rp));
        rp.integrator = PHIP_INTEGRATOR_VOLPATH_SIMPLE;
        rp.max_depth = m_maxDepth; rp.rr_depth = m_rrDepth;
        rp.strict_normals = m_strictNormals; rp.hide_emitters = m_hideEmitters;
        return m_holder.render(scene, queue, job, rp, "volpath_simple_hip");
    }

    void cancel() { if (m_holder.get()) phip_cancel(m_holder.get()); }

    MTS_DECLARE_CLASS()
private:
    PhipSceneHolder m_holder;
    ref<SamplingIntegrator> m_cpuPath;
};

MTS_IMPLEMENT_CLASS_S(VolPathSimpleHIP, false, MonteCarloIntegrator)
MTS_EXPORT_PLUGIN(VolPathSimpleHIP, "MI355X path tracer without MIS (volpath_simple_hip)");
MTS_NAMESPACE_END
