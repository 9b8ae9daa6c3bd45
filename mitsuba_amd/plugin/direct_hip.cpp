/*
 * direct_hip.cpp -- Mitsuba 0.6 integrator plugin `direct_hip`: MIDirectIntegrator (src/integrators/direct/direct.cpp) on the
 * MI355X back end.  Same shim as path_hip.cpp (which has the build recipe); select it with <integrator type="direct_hip"/>.
 *     plugins += env.SharedLibrary('direct_hip', ['path_hip/direct_hip.cpp'], LIBS = env['LIBS'] + ['phip'])
 * Parameters parse like `direct` (direct.cpp:91-108): shadingSamples, emitterSamples, bsdfSamples, strictNormals, hideEmitters.
 * Li() (used when the integrator is nested, direct.cpp:203-208) delegates to a CPU `direct` with the same parameters.
 */
#include "phip_flatten.h"

MTS_NAMESPACE_BEGIN

class DirectHIP : public SamplingIntegrator {
public:
    DirectHIP(const Properties &props) : SamplingIntegrator(props) {
        size_t shadingSamples = props.getSize("shadingSamples", 1);
        m_emitterSamples = props.getSize("emitterSamples", shadingSamples);
        m_bsdfSamples = props.getSize("bsdfSamples", shadingSamples);
        m_strictNormals = props.getBoolean("strictNormals", false);
        m_hideEmitters = props.getBoolean("hideEmitters", false);
        Assert(m_emitterSamples + m_bsdfSamples > 0);
        m_holder.setDevice(props.getInteger("device", 0));
        m_holder.setDeviceCount(props.getInteger("devices", 1));       /* GPUs of the node to spread the job over (0 = all) */
        Properties p("direct");
        p.setSize("emitterSamples", m_emitterSamples); p.setSize("bsdfSamples", m_bsdfSamples);
        p.setBoolean("strictNormals", m_strictNormals); p.setBoolean("hideEmitters", m_hideEmitters);
        m_cpuDirect = static_cast<SamplingIntegrator *>(PluginManager::getInstance()->createObject(MTS_CLASS(Integrator), p));
    }

    DirectHIP(Stream *stream, InstanceManager *manager) : SamplingIntegrator(stream, manager) {
        m_emitterSamples = stream->readSize(); m_bsdfSamples = stream->readSize();
        m_strictNormals = stream->readBool(); m_hideEmitters = stream->readBool();
        m_holder.setDevice(stream->readInt());
        m_holder.setDeviceCount(stream->readInt());
        m_cpuDirect = static_cast<SamplingIntegrator *>(manager->getInstance(stream));
    }

    void serialize(Stream *stream, InstanceManager *manager) const {
        SamplingIntegrator::serialize(stream, manager);
        stream->writeSize(m_emitterSamples); stream->writeSize(m_bsdfSamples);
        stream->writeBool(m_strictNormals); stream->writeBool(m_hideEmitters);
        stream->writeInt(m_holder.getDevice());
        stream->writeInt(m_holder.getDeviceCount());
        manager->serialize(stream, m_cpuDirect.get());
    }

    void configureSampler(const Scene *scene, Sampler *sampler) {
        /* the nested CPU integrator requests its sample arrays (direct.cpp:140-146); the device draws from the ctr stream */
        SamplingIntegrator::configureSampler(scene, sampler);
        m_cpuDirect->configureSampler(scene, sampler);
    }

    Spectrum Li(const RayDifferential &ray, RadianceQueryRecord &rRec) const {
        static bool told = false;      /* (only a wrapping integrator -- `adaptive`, `irrcache` -- gets here) */
        if (!told) { told = true; SLog(EWarn, "direct_hip: Li() was called by a wrapping integrator -- these samples run on the CPU (nested `direct`), not on the GPU"); }
        return m_cpuDirect->Li(ray, rRec);
    }

    bool preprocess(const Scene *scene, RenderQueue *queue, const RenderJob *job, int sceneResID, int sensorResID, int samplerResID) {
        if (!SamplingIntegrator::preprocess(scene, queue, job, sceneResID, sensorResID, samplerResID))
            return false;
        m_holder.flatten(scene);
        return true;
    }

    bool render(Scene *scene, RenderQueue *queue, const RenderJob *job, int sceneResID, int sensorResID, int samplerResID) {
        phip_render_params rp; memset(&rp, 0, sizeof(rp));
        rp.integrator = PHIP_INTEGRATOR_DIRECT;
        rp.emitter_samples = (int32_t) m_emitterSamples; rp.bsdf_samples = (int32_t) m_bsdfSamples;
        rp.max_depth = -1; rp.rr_depth = 5;                         /* unused by `direct` */
        rp.strict_normals = m_strictNormals; rp.hide_emitters = m_hideEmitters;
        return m_holder.render(scene, queue, job, rp, "direct_hip");
    }

    void cancel() { if (m_holder.get()) phip_cancel(m_holder.get()); }

    MTS_DECLARE_CLASS()
private:
    PhipSceneHolder m_holder;
    size_t m_emitterSamples, m_bsdfSamples;
    bool m_strictNormals, m_hideEmitters;
    ref<SamplingIntegrator> m_cpuDirect;
};

MTS_IMPLEMENT_CLASS_S(DirectHIP, false, SamplingIntegrator)
MTS_EXPORT_PLUGIN(DirectHIP, "MI355X direct illumination integrator (direct_hip)");
MTS_NAMESPACE_END
