/* ref_shims: minimal stand-ins for the Boost headers the reference's sources include, mapped onto the C++17 standard
 * library, so that the reference's own files under /root/reference compile in place (oracle/Makefile.ref).  Test
 * infrastructure only; nothing here is reference code. */
#pragma once
#define BOOST_VERSION 106000
