/* ORACLE -- TEST INFRASTRUCTURE ONLY.
 * xercesc/_mini.hpp -- the sliver of the Xerces-C 3 SAX1 interface that the reference's XML front end uses
 * (src/librender/scenehandler.cpp, src/librender/util.cpp, src/mitsuba/mitsuba.cpp), on top of a ~150-line non-validating XML reader:
 * elements, attributes, comments, processing instructions / the XML declaration, CDATA, the five predefined entities and numeric
 * character references.  Xerces-C itself is not in this image; with this header the reference's OWN SceneHandler and its OWN
 * command-line front end compile from /root/reference unchanged, so that `mitsuba scene.xml` runs end to end (oracle/Makefile.ref:
 * _ref/mitsuba).  What is NOT here: validation against data/schema/scene.xsd (the setters exist and do nothing -- a scene that the schema
 * would reject reaches SceneHandler's own checks instead), DTDs, namespaces, encodings other than UTF-8 / ASCII.
 * XMLCh is char: transcoding is the identity. */
#pragma once
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <stdexcept>

#define XERCES_CPP_NAMESPACE xercesc_mini
#define XERCES_CPP_NAMESPACE_BEGIN namespace xercesc_mini {
#define XERCES_CPP_NAMESPACE_END }
#define XERCES_CPP_NAMESPACE_USE using namespace xercesc_mini;
#define XERCES_CPP_NAMESPACE_QUALIFIER xercesc_mini::

/* (global typedefs, as in xercesc/util/XercesDefs.hpp) */
typedef char XMLCh;
typedef unsigned char XMLByte;
typedef size_t XMLSize_t;
typedef size_t XMLFileLoc;

namespace xercesc_mini {

struct XMLString {
    static XMLSize_t stringLen(const XMLCh *s) { return s ? strlen(s) : 0; }
    static XMLCh *transcode(const char *s) { char *r = new char[strlen(s) + 1]; strcpy(r, s); return r; }
    static void release(XMLCh **p) { delete[] *p; *p = nullptr; }
};

struct XMLTranscoder {
    enum UnRepOpts { UnRep_Throw, UnRep_RepChar };
    XMLSize_t transcodeTo(const XMLCh *src, XMLSize_t srcCount, XMLByte *dst, XMLSize_t maxBytes, XMLSize_t &charsEaten, UnRepOpts) {
        const XMLSize_t n = srcCount < maxBytes ? srcCount : maxBytes;
        memcpy(dst, src, n); charsEaten = n; return n;
    }
};
struct XMLTransService {
    enum Codes { Ok, UnsupportedEncoding, InternalFailure, SupportFilesNotFound };
    XMLTranscoder *makeNewTranscoderFor(const char *, Codes &c, XMLSize_t) { c = Ok; return new XMLTranscoder(); }
};
struct XMLPlatformUtils {
    static inline XMLTransService *fgTransService = new XMLTransService();
    static void Initialize() { }
    static void Terminate() { }
};

struct XMLException {
    std::string msg;
    explicit XMLException(const std::string &m = "") : msg(m) { }
    const XMLCh *getMessage() const { return msg.c_str(); }
};

struct Locator {
    XMLFileLoc line = 1, column = 1;
    std::string systemId;
    XMLFileLoc getLineNumber() const { return line; }
    XMLFileLoc getColumnNumber() const { return column; }
    const XMLCh *getSystemId() const { return systemId.c_str(); }
    const XMLCh *getPublicId() const { return ""; }
};

struct SAXException {
    std::string msg;
    explicit SAXException(const std::string &m = "") : msg(m) { }
    virtual ~SAXException() { }
    const XMLCh *getMessage() const { return msg.c_str(); }
};
struct SAXParseException : SAXException {
    XMLFileLoc line, column; std::string systemId;
    SAXParseException(const std::string &m, const Locator &l) : SAXException(m), line(l.line), column(l.column), systemId(l.systemId) { }
    XMLFileLoc getLineNumber() const { return line; }
    XMLFileLoc getColumnNumber() const { return column; }
    const XMLCh *getSystemId() const { return systemId.c_str(); }
};

struct AttributeList {
    std::vector<std::pair<std::string, std::string>> a;
    XMLSize_t getLength() const { return a.size(); }
    const XMLCh *getName(XMLSize_t i) const { return a[i].first.c_str(); }
    const XMLCh *getValue(XMLSize_t i) const { return a[i].second.c_str(); }
    const XMLCh *getType(XMLSize_t) const { return "CDATA"; }
    const XMLCh *getValue(const XMLCh *name) const { for (auto &p : a) if (p.first == name) return p.second.c_str(); return nullptr; }
};

struct DocumentHandler { virtual ~DocumentHandler() { } };
struct ErrorHandler { virtual ~ErrorHandler() { } };

/* SAX1 handler with no-op defaults, as xercesc::HandlerBase */
struct HandlerBase : DocumentHandler, ErrorHandler {
    virtual void characters(const XMLCh *const, const XMLSize_t) { }
    virtual void endDocument() { }
    virtual void endElement(const XMLCh *const) { }
    virtual void ignorableWhitespace(const XMLCh *const, const XMLSize_t) { }
    virtual void processingInstruction(const XMLCh *const, const XMLCh *const) { }
    virtual void resetDocument() { }
    virtual void setDocumentLocator(const Locator *const) { }
    virtual void startDocument() { }
    virtual void startElement(const XMLCh *const, AttributeList &) { }
    virtual void warning(const SAXParseException &) { }
    virtual void error(const SAXParseException &) { }
    virtual void fatalError(const SAXParseException &e) { throw e; }
    virtual void resetErrors() { }
};

struct MemBufInputSource {
    const XMLByte *data; XMLSize_t size; std::string name;
    MemBufInputSource(const XMLByte *d, XMLSize_t n, const XMLCh *id, bool = false) : data(d), size(n), name(id ? id : "") { }
};

class SAXParser {
public:
    enum ValSchemes { Val_Never, Val_Always, Val_Auto };
    void setValidationScheme(ValSchemes) { }                 /* (no schema validation here: see the header comment) */
    void setDoSchema(bool) { }
    void setDoNamespaces(bool) { }
    void setValidationSchemaFullChecking(bool) { }
    void setExternalNoNamespaceSchemaLocation(const char *) { }
    void setCalculateSrcOfs(bool) { }
    void setDocumentHandler(HandlerBase *h) { handler = h; }
    void setErrorHandler(HandlerBase *h) { errors = h; }
    XMLFileLoc getSrcOffset() const { return pos; }
    void parse(const char *filename) {
        FILE *f = fopen(filename, "rb");
        loc.systemId = filename;
        if (!f) { fail(std::string("unable to open the file '") + filename + "'"); return; }
        std::string s; char buf[65536]; size_t n;
        while ((n = fread(buf, 1, sizeof(buf), f)) > 0) s.append(buf, n);
        fclose(f);
        run(s);
    }
    void parse(const MemBufInputSource &src) { loc.systemId = src.name; run(std::string((const char *) src.data, src.size)); }

private:
    HandlerBase *handler = nullptr, *errors = nullptr;
    Locator loc; size_t pos = 0; const std::string *text = nullptr;

    void fail(const std::string &m) { SAXParseException e(m, loc); if (errors) errors->fatalError(e); else throw e; }
    void advance(size_t n) { for (size_t i = 0; i < n && pos < text->size(); ++i, ++pos) { if ((*text)[pos] == '\n') { ++loc.line; loc.column = 1; } else ++loc.column; } }
    bool at(const char *s) const { return text->compare(pos, strlen(s), s) == 0; }
    void skipWs() { while (pos < text->size() && strchr(" \t\r\n", (*text)[pos])) advance(1); }
    static bool nameChar(char c) { return isalnum((unsigned char) c) || c == '_' || c == '-' || c == '.' || c == ':' || (unsigned char) c >= 0x80; }
    std::string name() { const size_t b = pos; while (pos < text->size() && nameChar((*text)[pos])) advance(1); return text->substr(b, pos - b); }
    static void utf8(std::string &out, unsigned long cp) {
        if (cp < 0x80) out += (char) cp;
        else if (cp < 0x800) { out += (char) (0xC0 | (cp >> 6)); out += (char) (0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { out += (char) (0xE0 | (cp >> 12)); out += (char) (0x80 | ((cp >> 6) & 0x3F)); out += (char) (0x80 | (cp & 0x3F)); }
        else { out += (char) (0xF0 | (cp >> 18)); out += (char) (0x80 | ((cp >> 12) & 0x3F)); out += (char) (0x80 | ((cp >> 6) & 0x3F)); out += (char) (0x80 | (cp & 0x3F)); }
    }
    std::string unescape(const std::string &s) {
        std::string out; out.reserve(s.size());
        for (size_t i = 0; i < s.size(); ++i) {
            if (s[i] != '&') { out += s[i]; continue; }
            const size_t e = s.find(';', i);
            if (e == std::string::npos) { fail("unterminated entity reference"); return out; }
            const std::string ent = s.substr(i + 1, e - i - 1);
            if (ent == "lt") out += '<'; else if (ent == "gt") out += '>'; else if (ent == "amp") out += '&';
            else if (ent == "quot") out += '"'; else if (ent == "apos") out += '\'';
            else if (!ent.empty() && ent[0] == '#') utf8(out, ent.size() > 1 && (ent[1] == 'x' || ent[1] == 'X') ? strtoul(ent.c_str() + 2, nullptr, 16) : strtoul(ent.c_str() + 1, nullptr, 10));
            else { fail("unknown entity '&" + ent + ";'"); return out; }
            i = e;
        }
        return out;
    }
    void run(const std::string &s) {
        text = &s; pos = 0; loc.line = 1; loc.column = 1;
        if (s.compare(0, 3, "\xEF\xBB\xBF") == 0) pos = 3;
        if (!handler) return;
        handler->setDocumentLocator(&loc);
        handler->startDocument();
        std::vector<std::string> open;
        bool sawRoot = false;
        while (pos < s.size()) {
            if (s[pos] != '<') {
                const size_t b = pos; while (pos < s.size() && s[pos] != '<') advance(1);
                const std::string chars = unescape(s.substr(b, pos - b));
                if (!open.empty()) handler->characters(chars.c_str(), chars.size());
                else if (chars.find_first_not_of(" \t\r\n") != std::string::npos) { fail("character data outside of the root element"); return; }
                continue;
            }
            if (at("<!--")) { const size_t e = s.find("-->", pos + 4); if (e == std::string::npos) { fail("unterminated comment"); return; } advance(e + 3 - pos); continue; }
            if (at("<?")) { const size_t e = s.find("?>", pos + 2); if (e == std::string::npos) { fail("unterminated processing instruction"); return; } advance(e + 2 - pos); continue; }
            if (at("<![CDATA[")) {
                const size_t e = s.find("]]>", pos + 9); if (e == std::string::npos) { fail("unterminated CDATA section"); return; }
                const std::string chars = s.substr(pos + 9, e - pos - 9);
                if (!open.empty()) handler->characters(chars.c_str(), chars.size());
                advance(e + 3 - pos); continue;
            }
            if (at("<!")) { const size_t e = s.find('>', pos); if (e == std::string::npos) { fail("unterminated declaration"); return; } advance(e + 1 - pos); continue; }   /* DOCTYPE: skipped */
            if (at("</")) {
                advance(2); const std::string n = name(); skipWs();
                if (pos >= s.size() || s[pos] != '>') { fail("malformed end tag </" + n + ">"); return; }
                advance(1);
                if (open.empty() || open.back() != n) { fail("end tag </" + n + "> does not match" + (open.empty() ? std::string(" any open element") : " <" + open.back() + ">")); return; }
                open.pop_back();
                handler->endElement(n.c_str());
                continue;
            }
            advance(1);
            const std::string n = name();
            if (n.empty()) { fail("malformed start tag"); return; }
            if (open.empty() && sawRoot) { fail("more than one root element"); return; }
            sawRoot = true;
            AttributeList attrs;
            bool selfClosing = false;
            for (;;) {
                skipWs();
                if (pos >= s.size()) { fail("unterminated start tag <" + n + ">"); return; }
                if (s[pos] == '>') { advance(1); break; }
                if (at("/>")) { advance(2); selfClosing = true; break; }
                const std::string an = name();
                if (an.empty()) { fail("malformed attribute in <" + n + ">"); return; }
                skipWs();
                if (pos >= s.size() || s[pos] != '=') { fail("attribute '" + an + "' of <" + n + "> has no value"); return; }
                advance(1); skipWs();
                if (pos >= s.size() || (s[pos] != '"' && s[pos] != '\'')) { fail("the value of attribute '" + an + "' must be quoted"); return; }
                const char q = s[pos]; advance(1);
                const size_t b = pos; while (pos < s.size() && s[pos] != q) advance(1);
                if (pos >= s.size()) { fail("unterminated attribute value"); return; }
                for (auto &p : attrs.a) if (p.first == an) { fail("attribute '" + an + "' appears twice in <" + n + ">"); return; }
                attrs.a.push_back({ an, unescape(s.substr(b, pos - b)) });
                advance(1);
            }
            handler->startElement(n.c_str(), attrs);
            if (selfClosing) handler->endElement(n.c_str()); else open.push_back(n);
        }
        if (!open.empty()) { fail("unexpected end of the document: <" + open.back() + "> is still open"); return; }
        if (!sawRoot) { fail("the document has no root element"); return; }
        handler->endDocument();
    }
};

} // namespace xercesc_mini
